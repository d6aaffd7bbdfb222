#!/bin/bash
# round 5, call 6: transforms resident on a CU subset beside a capped, token-chained GEMM of the other stream
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "conv_winograd" 2>&1 | tail -4 ) > $O/r5f_checks.log 2>&1
echo "== checks"; cat $O/r5f_checks.log
( PF_W3_TGRID=64 timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "conv_winograd" 2>&1 | tail -4 ) > $O/r5f_checks_resident.log 2>&1
echo "== checks resident"; cat $O/r5f_checks_resident.log
( timeout 900 python tools/overlap_probe.py 6 ) > $O/r5f_overlap_probe.md 2>&1
echo "== overlap"; cat $O/r5f_overlap_probe.md
