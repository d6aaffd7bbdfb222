#!/bin/bash
# round-2 GPU call 1: facts first (fp32 headline, precision probe, experimental GEMM validation, per-layer sweeps)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( PF_TEST_EXPERIMENTAL=1 timeout 400 python -m pytest tests/test_experimental_gpu.py -m gpu -q -x 2>&1 | tail -15 ) > gpurun_out/r2_experimental.log 2>&1
( timeout 500 python tools/precision_probe.py ) > gpurun_out/r2_precision_probe.log 2>&1
( timeout 400 python bench.py --steps 3 --warmup 1 ) > gpurun_out/r2_bench1.json 2> gpurun_out/r2_bench1.err
( timeout 200 python bench.py --gemm-sweep --dtype fp32 ) > gpurun_out/r2_sweep_fp32.log 2>&1
( timeout 100 python bench.py --gemm-sweep --dtype bf16 ) > gpurun_out/r2_sweep_bf16.log 2>&1
( PF_GEMM_PERSIST=1 timeout 100 python bench.py --gemm-sweep --dtype bf16 ) > gpurun_out/r2_sweep_bf16_persist.log 2>&1
tail -3 gpurun_out/r2_experimental.log; tail -12 gpurun_out/r2_precision_probe.log; cat gpurun_out/r2_bench1.json; cat gpurun_out/r2_sweep_fp32.log
