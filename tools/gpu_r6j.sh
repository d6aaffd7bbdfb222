#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_attndbg.so timeout 300 python tools/attn_blocks.py 8 ) > gpurun_out/r6j_attn_blocks.log 2>&1; grep -v amdgpu.ids gpurun_out/r6j_attn_blocks.log
