#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python tools/f32_tune.py $O/r2_f32_tune_256.json ) > $O/r2_f32_tune_256.log 2>&1
( timeout 600 python -m pytest tests/test_hip_ops_gpu.py tests/test_persistent_gemm_gpu.py -m gpu -q -x --durations=8 -k "dominant or persistent or big_gemm or PIPE" 2>&1 | tail -16 ) > $O/r2c11_slowtests.log 2>&1
( PF_IGEMM_CFG=7 timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "fp32 and conv and not dominant" 2>&1 | tail -3 ) > $O/r2c11_cfg7_checks.log 2>&1
cat $O/r2_f32_tune_256.log; cat $O/r2c11_slowtests.log; tail -n 3 $O/r2c11_cfg7_checks.log
