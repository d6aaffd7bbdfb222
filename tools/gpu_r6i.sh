#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_attndbg.so timeout 300 python tools/attn_timeline.py ) > gpurun_out/r6i_attn_timeline.log 2>&1; grep -v amdgpu.ids gpurun_out/r6i_attn_timeline.log
