#!/bin/bash
# round 5, final session on the final tree: measurements (tools/gpu_final_r5.sh) + the whole GPU suite + smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
bash tools/gpu_final_r5.sh
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -20 ) > gpurun_out/r5_pytest_gpu.log 2>&1
cat gpurun_out/r5_pytest_gpu.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > gpurun_out/r5_smoke.log 2>&1
cat gpurun_out/r5_smoke.log
