#!/bin/bash
# round 5, call 8: exchanged non-bare epilogue (whole-line f32 stores, 16-byte plane stores) of the 192-tile kernel
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "gemm_split3 or conv_winograd or vit" 2>&1 | tail -8 ) > $O/r5h_checks.log 2>&1
echo "== checks"; cat $O/r5h_checks.log
( PF_S3_T192=2 timeout 600 python tools/persist_probe.py envsweep:PF_S3_FLAGS=0,2 ) > $O/r5h_flags_sweep.md 2>&1
echo "== flags sweep"; cat $O/r5h_flags_sweep.md
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "PF_S3_FLAGS=0" "PF_S3_FLAGS=2" ) > $O/r5h_image_ab.md 2> $O/r5h_image_ab.err
echo "== image ab"; cat $O/r5h_image_ab.md; tail -2 $O/r5h_image_ab.err
