#!/bin/bash
# session S: split GEMM 3-stage ring vs 2-stage
mkdir -p gpurun_out
timeout 300 python tools/gpu_selfcheck.py gemm_split3 > gpurun_out/r3s_check.log 2>&1
tail -3 gpurun_out/r3s_check.log | cut -c1-900
for ns in 3 2; do
  echo "== stages $ns"
  PF_S3_STAGES=$ns timeout 300 python tools/split3_probe.py 8296 > gpurun_out/r3s_probe_ns$ns.md 2>&1
  cat gpurun_out/r3s_probe_ns$ns.md
done
