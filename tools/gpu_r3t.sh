#!/bin/bash
# session T: split GEMM with register-double-buffered fragments
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py gemm_split3 > gpurun_out/r3t_check.log 2>&1
tail -2 gpurun_out/r3t_check.log | cut -c1-300
timeout 300 python tools/split3_probe.py 8296 4148 > gpurun_out/r3t_probe.md 2>&1
cat gpurun_out/r3t_probe.md
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > gpurun_out/r3t_bench.json 2> gpurun_out/r3t_bench.err
tail -1 gpurun_out/r3t_bench.json | cut -c1-330
