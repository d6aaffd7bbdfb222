#!/bin/bash
# session Q: split-precision GEMM correctness + speed
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py gemm_split3 > gpurun_out/r3q_check.log 2>&1
tail -5 gpurun_out/r3q_check.log
timeout 300 python tools/split3_probe.py > gpurun_out/r3q_split3_probe.md 2>&1
cat gpurun_out/r3q_split3_probe.md
