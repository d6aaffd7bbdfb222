#!/bin/bash
# round 5, call 9: the single exchanged epilogue of the 192-tile kernel (no fallback path: 222 / 230 VGPRs, no scratch)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "gemm_split3 or conv_winograd or vit" 2>&1 | tail -8 ) > $O/r5i_checks.log 2>&1
echo "== checks"; cat $O/r5i_checks.log
( PF_S3_T192=2 timeout 600 python tools/persist_probe.py envsweep:PF_S3_BLOAD=0,1 ) > $O/r5i_sweep.md 2>&1
echo "== sweep"; cat $O/r5i_sweep.md
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "PF_S3_BLOAD=0,PF_S3_ORDER=1" "" ) > $O/r5i_image_ab.md 2> $O/r5i_image_ab.err
echo "== image ab"; cat $O/r5i_image_ab.md; tail -2 $O/r5i_image_ab.err
