#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -k "vit_attention" -s 2>&1 | tail -12 ) > gpurun_out/r6l_checks.log 2>&1; tail -c 2500 gpurun_out/r6l_checks.log
( timeout 900 python tools/image_ab.py --steps 4 --rounds 3 "PF_ATTN_V2=0" "" "PF_ATTN_SCHED=1" ) > gpurun_out/r6l_image_ab_attention.md 2> gpurun_out/r6l_image_ab.err; cat gpurun_out/r6l_image_ab_attention.md; tail -3 gpurun_out/r6l_image_ab.err
