#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py conv_bf16_pp conv_big_gemm_gelu_k1024 > gpurun_out/r3ah_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3ah_check.log | cut -c1-420
for m in 1 0; do
  echo "== PF_BF16_PP=$m"
  PF_BF16_PP=$m timeout 200 python bench.py --gemm-sweep --dtype bf16 --only qkv,proj,fc1,fc2 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3ah_sweep.log
for m in 1 0; do
  PF_BF16_PP=$m timeout 300 python bench.py --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bf16 PF_BF16_PP=$m', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3ah_bench.log
done
