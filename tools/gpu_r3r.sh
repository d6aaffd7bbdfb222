#!/bin/bash
# session R: split-precision ViT linears e2e: producers check, bench A/B, headline parity in the new default
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py gemm_split3 > gpurun_out/r3r_check.log 2>&1
tail -3 gpurun_out/r3r_check.log
for m in 1 0; do
  PF_LINEAR_SPLIT3=$m timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/r3r_bench_split$m.json 2> gpurun_out/r3r_bench_split$m.err
  tail -1 gpurun_out/r3r_bench_split$m.json | cut -c1-400
done
timeout 900 python -m pytest tests/test_headline_parity_gpu.py tests/test_e2e_gpu.py -m gpu -x -q > gpurun_out/r3r_pytest.log 2>&1
tail -5 gpurun_out/r3r_pytest.log
