#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py conv_winograd conv_winograd_fused > gpurun_out/r3ai_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3ai_check.log | cut -c1-260
timeout 400 python tools/wino_fused_probe.py time c544_544,c768_768_L4,c768_256_L4,c512_256_L4,c768_768_L3,c256_256_L4,c768_768_L2,c512_256_L2,c768_768_L1,c256_256_B1 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee gpurun_out/r3ai_three_step_split.log
