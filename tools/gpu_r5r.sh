#!/bin/bash
# round 5, call 18: narrow last column tile of the 192-tile kernel (N = 544: 5 fragments per wave column instead of 6 + 4 live of 6)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "gemm_split3 or conv_winograd or vit" 2>&1 | tail -6 ) > $O/r5r_checks.log 2>&1
echo "== checks"; cat $O/r5r_checks.log
( PF_S3_T192=2 timeout 600 python tools/persist_probe.py envsweep:PF_S3_FLAGS=0,8 ) > $O/r5r_narrow_sweep.md 2>&1
echo "== sweep"; cat $O/r5r_narrow_sweep.md
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "PF_S3_FLAGS=0" "PF_S3_FLAGS=8" ) > $O/r5r_image_ab.md 2> $O/r5r_image_ab.err
echo "== image ab"; cat $O/r5r_image_ab.md; tail -2 $O/r5r_image_ab.err
