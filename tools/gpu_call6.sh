#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python tools/f32_tune.py $O/r2_f32_tune_ws.json ) > $O/r2_f32_tune_ws.log 2>&1
( timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "fp32 and conv and not dominant" 2>&1 | tail -4 ) > $O/r2c6_opchecks_ws.log 2>&1
( timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary ) > $O/r2c6_bench_ws.json 2> $O/r2c6_bench_ws.err
( timeout 300 python -m pytest tests/test_e2e_gpu.py -m gpu -q -x -k "fp32 or golden" 2>&1 | tail -4 ) > $O/r2c6_e2e.log 2>&1
cat $O/r2_f32_tune_ws.log; tail -n 3 $O/r2c6_opchecks_ws.log; cat $O/r2c6_bench_ws.json; tail -n 3 $O/r2c6_e2e.log
