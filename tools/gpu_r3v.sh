#!/bin/bash
mkdir -p gpurun_out
PF_LIB_PATH=patchfusion_amd/libpf_wfdbg.so timeout 200 python tools/split3_decomp.py 8296 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3v_decomp.log
