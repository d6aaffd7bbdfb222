#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
S=c544_544,c768_768_L4,c768_256_L4,c544_32,c512_256_L4,c256_256_L4,c768_768_L3
for v in a b hip a b hip; do
  L=$PWD/patchfusion_amd/libpf_wf_$v.so; [ $v = hip ] && L=$PWD/patchfusion_amd/libpf_hip.so
  echo "=== variant $v"
  ( PF_LIB_PATH=$L timeout 300 python tools/wino_fused_probe.py time $S ) 2>&1 | cut -c1-48,118-140,330-420 | grep -v amdgpu.ids
done > gpurun_out/r3j_variants.log 2>&1
cat gpurun_out/r3j_variants.log
