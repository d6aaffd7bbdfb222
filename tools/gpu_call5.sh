#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( PF_IGEMM_CFG=1 timeout 100 python tools/abl_run.py fp32 ) > $O/r2_abl_f32.log 2>&1
for v in nodma nobar nolds nodma_nobar mfmaonly; do ( PF_LIB_PATH=$PWD/gpurun_abl/libpf_abl_$v.so PF_IGEMM_CFG=1 timeout 100 python tools/abl_run.py fp32 ) >> $O/r2_abl_f32.log 2>&1; done
grep -v amdgpu $O/r2_abl_f32.log
