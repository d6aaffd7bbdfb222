#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
export PF_LIB_PATH=$PWD/patchfusion_amd/libpf_attndbg.so
for lds in 73728 90000; do
  echo "== dynamic LDS $lds"
  PF_ATTN_LDS=$lds timeout 300 python tools/attn_split3_time.py 3 2>&1 | grep -v amdgpu.ids
done > gpurun_out/r6c_attn_one_block_per_cu.log 2>&1
cat gpurun_out/r6c_attn_one_block_per_cu.log
