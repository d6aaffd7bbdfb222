#!/bin/bash
# round 5, call 2: group B of the 192-tile kernel issues its W pieces from its own load phase (PF_S3_BLOAD=1, default) vs between its MFMAs (0)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "gemm_split3 or conv_winograd" 2>&1 | tail -8 ) > $O/r5b_checks.log 2>&1
echo "== checks"; cat $O/r5b_checks.log
( PF_S3_T192=2 timeout 600 python tools/persist_probe.py envsweep:PF_S3_BLOAD=0,1 ) > $O/r5b_bload_sweep.md 2>&1
echo "== sweep"; cat $O/r5b_bload_sweep.md
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "PF_S3_BLOAD=0" "PF_S3_BLOAD=1" "PF_S3_BLOAD=1,PF_S3_ORDER=2" ) > $O/r5b_image_ab.md 2> $O/r5b_image_ab.err
echo "== image ab"; cat $O/r5b_image_ab.md; tail -3 $O/r5b_image_ab.err
