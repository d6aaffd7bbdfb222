#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
PF_ATTN_SCHED=2 bash tools/attn_pmc.sh r6r 32 8 > gpurun_out/r6r_attn_pmc.log 2>&1; tail -25 gpurun_out/r6r_attn_pmc.log
( timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -20 ) > gpurun_out/r6_pytest_gpu_mid.log 2>&1; cat gpurun_out/r6_pytest_gpu_mid.log
