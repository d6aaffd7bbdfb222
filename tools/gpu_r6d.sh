#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "vit_attention_split3" -s 2>&1 | tail -30 ) > gpurun_out/r6d_checks.log 2>&1; tail -c 3000 gpurun_out/r6d_checks.log
( timeout 300 python tools/attn_split3_time.py 3 ) > gpurun_out/r6d_attn_time.log 2>&1; grep -v amdgpu.ids gpurun_out/r6d_attn_time.log
