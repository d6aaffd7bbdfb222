#!/bin/bash
# round 5, call 10: workspace cap (sub-batched three-step layers), free_parameters(), where the memory goes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "conv_winograd" 2>&1 | tail -8 ) > $O/r5j_checks.log 2>&1
echo "== checks"; cat $O/r5j_checks.log
( timeout 600 python tools/mem_probe.py ) > $O/r5j_mem_probe.md 2>&1
echo "== mem"; cat $O/r5j_mem_probe.md
( PF_WS_CAP_GB=100 timeout 600 python tools/mem_probe.py ) > $O/r5j_mem_probe_nocap.md 2>&1
echo "== mem nocap"; tail -5 $O/r5j_mem_probe_nocap.md
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "PF_WS_CAP_GB=100" "PF_WS_CAP_GB=10" "PF_WS_CAP_GB=5" ) > $O/r5j_image_ab.md 2> $O/r5j_image_ab.err
echo "== image ab"; cat $O/r5j_image_ab.md; tail -2 $O/r5j_image_ab.err
