#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/wino_fused_probe.py time c64_32 c32_32 c32_256 c64_64_L4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ag_small_layers.log
timeout 300 python tools/gpu_selfcheck.py conv_winograd_fused conv_winograd conv3x3_rcu conv3x3_nobias_48 > gpurun_out/r3ag_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3ag_check.log | cut -c1-200
for m in 1 0; do
  PF_WINO_FUSED_SMALL=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PF_WINO_FUSED_SMALL=$m', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3ag_bench.log
done
