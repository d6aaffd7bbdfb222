#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 60 ./tools/tr_probe ) > gpurun_out/r6a_tr_probe.log 2>&1; tail -5 gpurun_out/r6a_tr_probe.log
( timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "vit_attention_split3" -s 2>&1 | tail -30 ) > gpurun_out/r6a_checks.log 2>&1; tail -c 2500 gpurun_out/r6a_checks.log
( timeout 300 python tools/attn_split3_time.py 3 ) > gpurun_out/r6a_attn_time.log 2>&1; cat gpurun_out/r6a_attn_time.log
