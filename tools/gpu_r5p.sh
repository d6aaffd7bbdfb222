#!/bin/bash
# round 5, call 16: the whole GPU suite on the current tree
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -25 ) > $O/r5p_pytest_gpu.log 2>&1
cat $O/r5p_pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > $O/r5p_smoke.log 2>&1
cat $O/r5p_smoke.log
