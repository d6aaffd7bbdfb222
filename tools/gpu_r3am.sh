#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py vit_attention_split3 gemm_split3 layernorm conv_winograd > gpurun_out/r3am_check.log 2>&1
grep -E "PASS|FAIL|Error|error" gpurun_out/r3am_check.log | cut -c1-330
timeout 100 python tools/attn_probe.py split3 20 2>&1 | grep vit_attention | tee gpurun_out/r3am_attn.log
timeout 100 python tools/attn_probe.py fp32 20 2>&1 | grep vit_attention | tee -a gpurun_out/r3am_attn.log
for m in 1 0; do
  PF_ATTN_SPLIT3=$m timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PF_ATTN_SPLIT3=$m', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3am_bench.log
done
