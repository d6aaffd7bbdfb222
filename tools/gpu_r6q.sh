#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -k "vit_attention" 2>&1 | tail -6 ) > gpurun_out/r6q_checks.log 2>&1; tail -c 1200 gpurun_out/r6q_checks.log
( timeout 300 python tools/attn_split3_time.py 3 ) > gpurun_out/r6q_attn_time.log 2>&1; grep -v amdgpu.ids gpurun_out/r6q_attn_time.log
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_attndbg.so timeout 300 python tools/attn_blocks.py 8 ) > gpurun_out/r6q_attn_blocks.log 2>&1; grep -v amdgpu.ids gpurun_out/r6q_attn_blocks.log | head -8
