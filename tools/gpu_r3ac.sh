#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py bins_tail bins_ops > gpurun_out/r3ac_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3ac_check.log | cut -c1-300
timeout 200 python tools/bins_tail_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3ac_bins_tail.log
for m in 1 0; do
  PF_BINS_TAIL=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bins_tail=$m', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3ac_bench.log
done
