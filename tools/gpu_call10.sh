#!/bin/bash
# intermediate full GPU verification
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -40 ) > $O/r2c10_pytest_gpu.log 2>&1
( timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $O/r2c10_smoke.log 2>&1
cat $O/r2c10_pytest_gpu.log; tail -n 3 $O/r2c10_smoke.log
