#!/bin/bash
# round-3 GPU session G: fused Winograd v5 (five DMA pieces per M0 write)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 400 python tools/wino_fused_probe.py check ) > $O/r3g_wf_check.log 2>&1
tail -2 $O/r3g_wf_check.log
if grep -q "^0 failing cases" $O/r3g_wf_check.log; then
  ( timeout 500 python tools/wino_fused_probe.py time ) > $O/r3g_wf_time.log 2>&1
  cut -c1-48,118-420 $O/r3g_wf_time.log
  ( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_wfdbg.so timeout 300 python tools/wino_fused_probe.py timeline c544_544 ) > $O/r3g_timeline.log 2>&1
  grep -A8 "block 5000" $O/r3g_timeline.log | head -12; grep "mean interval\|last chunk" $O/r3g_timeline.log
fi
