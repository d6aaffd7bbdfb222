#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python tools/f32_tune.py $O/r2_f32_tune_ws3.json ) > $O/r2_f32_tune_ws3.log 2>&1
cat $O/r2_f32_tune_ws3.log
