#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_attndbg.so timeout 300 python tools/attn_v2_probe.py decomp ) > gpurun_out/r6b_attn_decomp.log 2>&1; cat gpurun_out/r6b_attn_decomp.log
bash tools/attn_pmc.sh r6b 2>&1 | tail -40
