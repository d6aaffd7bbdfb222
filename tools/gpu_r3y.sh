#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py resize_ops > gpurun_out/r3y_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3y_check.log | cut -c1-250
for v in "PF_RESIZE_DBG=0 PF_RESIZE_GX=64" "PF_RESIZE_DBG=1 PF_RESIZE_GX=64" "PF_RESIZE_DBG=2 PF_RESIZE_GX=64" "PF_RESIZE_DBG=0 PF_RESIZE_GX=16" "PF_RESIZE_DBG=0 PF_RESIZE_GX=4"; do
  echo "== $v"
  env $v timeout 200 python tools/resize_probe.py fp32 2>&1 | grep -v amdgpu.ids | cut -c1-90
done | tee gpurun_out/r3y_resize_variants.log
