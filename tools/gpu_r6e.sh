#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 300 python tools/attn_debug.py ) > gpurun_out/r6e_attn_debug.log 2>&1; grep -v amdgpu.ids gpurun_out/r6e_attn_debug.log
