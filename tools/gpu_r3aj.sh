#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py conv_winograd conv_winograd_fused gemm_split3 conv_dominant_launch > gpurun_out/r3aj_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3aj_check.log | cut -c1-200
timeout 300 python tools/wino_fused_probe.py time c544_544,c768_768_L4,c768_768_L3 2>&1 | grep -v amdgpu.ids | cut -c1-175 | tee gpurun_out/r3aj_three_step_split.log
timeout 200 python bench.py --roofline-only 2>/dev/null | tail -1 | cut -c1-900
for m in 1 0; do
  PF_WINO_SPLIT3=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PF_WINO_SPLIT3=$m', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3aj_bench.log
done
