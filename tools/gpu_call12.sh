#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "attention or conv" 2>&1 | tail -4 ) > $O/r2c12_checks.log 2>&1
( timeout 300 python -m pytest tests/test_e2e_gpu.py tests/test_headline_parity_gpu.py -m gpu -q -x 2>&1 | tail -4 ) > $O/r2c12_e2e.log 2>&1
( timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $O/r2c12_bench.json 2> $O/r2c12_bench.err
tail -n 3 $O/r2c12_checks.log $O/r2c12_e2e.log; cat $O/r2c12_bench.json
