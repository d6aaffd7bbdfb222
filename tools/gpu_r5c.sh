#!/bin/bash
# round 5, call 3: s_memtime timeline of the 192-tile kernel (debug build)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( PF_LIB_PATH=patchfusion_amd/libpf_wfdbg.so timeout 300 python tools/persist_probe.py timeline192 ) > $O/r5c_timeline192.md 2>&1
echo "== timeline"; cat $O/r5c_timeline192.md
