#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python tools/attn_debug.py ) > gpurun_out/r6k_attn_debug.log 2>&1; grep -v amdgpu.ids gpurun_out/r6k_attn_debug.log | grep -v "pipe - two\|by d block"
( timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -k "vit_attention_split3" -s 2>&1 | tail -30 ) > gpurun_out/r6k_checks.log 2>&1; tail -c 1500 gpurun_out/r6k_checks.log
( timeout 300 python tools/attn_split3_time.py 3 ) > gpurun_out/r6k_attn_time.log 2>&1; grep -v amdgpu.ids gpurun_out/r6k_attn_time.log
