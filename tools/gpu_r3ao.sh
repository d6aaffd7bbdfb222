#!/bin/bash
mkdir -p gpurun_out
timeout 200 python tools/gpu_selfcheck.py gemm_split3 > gpurun_out/r3ao_check.log 2>&1
grep -E "PASS|FAIL" gpurun_out/r3ao_check.log | cut -c1-120
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -1
