#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_wfdbg.so timeout 300 python tools/wino_fused_probe.py timeline c544_544 ) > gpurun_out/r3h_timeline.log 2>&1
cat gpurun_out/r3h_timeline.log
