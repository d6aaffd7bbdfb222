#!/bin/bash
# round-3 GPU session E: fused Winograd timeline (s_memtime, debug build), decomposition, T-wave ring depth A/B
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_wfdbg.so timeout 300 python tools/wino_fused_probe.py timeline c544_544 ) > $O/r3e_timeline.log 2>&1
cat $O/r3e_timeline.log
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_wfdbg.so timeout 300 python tools/wino_fused_probe.py decomp c544_544 ) > $O/r3e_decomp.log 2>&1
cat $O/r3e_decomp.log
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_wft9.so timeout 300 python tools/wino_fused_probe.py check ) > $O/r3e_t9_check.log 2>&1
tail -2 $O/r3e_t9_check.log
( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_wft9.so timeout 300 python tools/wino_fused_probe.py time c544_544,c768_768_L4,c544_32,c256_256_L4,c768_768_L2 ) > $O/r3e_t9_time.log 2>&1
cut -c1-48,118-420 $O/r3e_t9_time.log
