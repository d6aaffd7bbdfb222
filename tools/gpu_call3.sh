#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python tools/f32_tune.py $O/r2_f32_tune.json ) > $O/r2_f32_tune.log 2>&1
( PF_F32_PIPE=2 timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "fp32 and conv and not dominant" 2>&1 | tail -4 ) > $O/r2c3_opchecks_pipe2.log 2>&1
( PF_F32_PIPE=1 timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "fp32 and conv and not dominant" 2>&1 | tail -4 ) > $O/r2c3_opchecks_pipe1.log 2>&1
( timeout 400 python -m pytest tests/test_headline_parity_gpu.py -m gpu -q -x 2>&1 | tail -4 ) > $O/r2c3_headline.log 2>&1
cat $O/r2_f32_tune.log; tail -3 $O/r2c3_opchecks_pipe2.log $O/r2c3_opchecks_pipe1.log $O/r2c3_headline.log
