#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python tools/f32_tune.py $O/r2_f32_tune_pipe3.json ) > $O/r2_f32_tune_pipe3.log 2>&1
( PF_F32_PIPE=3 timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "fp32 and conv and not dominant" 2>&1 | tail -4 ) > $O/r2c4_opchecks_pipe3.log 2>&1
( PF_F32_PIPE=3 timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary ) > $O/r2c4_bench_pipe3.json 2> $O/r2c4_bench_pipe3.err
cat $O/r2_f32_tune_pipe3.log; tail -n 3 $O/r2c4_opchecks_pipe3.log; cat $O/r2c4_bench_pipe3.json
