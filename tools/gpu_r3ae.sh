#!/bin/bash
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py gemm_split3 > gpurun_out/r3ae_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3ae_check.log | cut -c1-400
for pp in 1 0; do
  echo "== PF_S3_PP=$pp"
  PF_S3_PP=$pp timeout 300 python tools/split3_probe.py 8296 4148 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r3ae_probe.log
