#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py vit_attention_split3 > gpurun_out/r3al_check.log 2>&1
grep -E "PASS|FAIL|Error|error" gpurun_out/r3al_check.log | cut -c1-500
timeout 100 python tools/attn_probe.py split3 20 2>&1 | grep vit_attention | tee gpurun_out/r3al_attn.log
timeout 100 python tools/attn_probe.py fp32 20 2>&1 | grep vit_attention | tee -a gpurun_out/r3al_attn.log
