#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -k "gemm_split3 or conv_split or conv_gemm_vitl" 2>&1 | tail -5 ) > gpurun_out/r6o_checks.log 2>&1; tail -c 1500 gpurun_out/r6o_checks.log
( timeout 600 python tools/vit_linear_time.py "PF_S3_BALANCE=0" "" ) > gpurun_out/r6o_balanced_walk.md 2>&1; grep -v amdgpu.ids gpurun_out/r6o_balanced_walk.md
( timeout 900 python tools/image_ab.py --steps 4 --rounds 3 "PF_S3_BALANCE=0" "" ) > gpurun_out/r6o_image_ab_balance.md 2> gpurun_out/r6o_image_ab.err; cat gpurun_out/r6o_image_ab_balance.md; tail -2 gpurun_out/r6o_image_ab.err
