#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 900 python tools/image_ab.py --steps 4 --rounds 3 "" "PF_VIT_BATCH_ALL=1" "PF_STREAMS=1" ) > gpurun_out/r6p_image_ab_vit_batch_all.md 2> gpurun_out/r6p_image_ab.err; cat gpurun_out/r6p_image_ab_vit_batch_all.md; tail -2 gpurun_out/r6p_image_ab.err
