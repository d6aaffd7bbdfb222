#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py resize_ops > gpurun_out/r3x_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3x_check.log | cut -c1-250
timeout 200 python tools/resize_probe.py fp32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3x_resize_fp32.log
timeout 200 python tools/resize_probe.py bf16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3x_resize_bf16.log
