#!/bin/bash
# round 5, call 15: tile windows below one image -- does the V / M arena pair stay in the 256 MB memory-side cache?
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "conv_winograd" 2>&1 | tail -8 ) > $O/r5o_checks.log 2>&1
echo "== checks"; cat $O/r5o_checks.log
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "PF_WS_CAP_GB=2.5" "PF_WS_CAP_GB=1.2" "PF_WS_CAP_GB=0.6" "PF_WS_CAP_GB=0.3" "PF_WS_CAP_GB=0.15" "PF_WS_CAP_GB=0.08" ) > $O/r5o_image_ab.md 2> $O/r5o_image_ab.err
echo "== image ab"; cat $O/r5o_image_ab.md; tail -2 $O/r5o_image_ab.err
