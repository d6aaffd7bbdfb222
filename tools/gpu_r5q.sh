#!/bin/bash
# round 5, call 17: split attention with register prefetch of the next key tile (PF_ATTN_PREF=1, two blocks per CU)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( PF_ATTN_PREF=1 timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "vit_attention" 2>&1 | tail -4 ) > $O/r5q_checks.log 2>&1
echo "== checks (PREF=1)"; cat $O/r5q_checks.log
for pr in 0 1 0 1; do echo "PF_ATTN_PREF=$pr"; PF_ATTN_PREF=$pr timeout 300 python tools/attn_split3_time.py 2>&1 | grep vit_attention; done > $O/r5q_attn_time.log 2>&1
echo "== time"; cat $O/r5q_attn_time.log
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "PF_ATTN_PREF=0" "PF_ATTN_PREF=1" ) > $O/r5q_image_ab.md 2> $O/r5q_image_ab.err
echo "== image ab"; cat $O/r5q_image_ab.md; tail -2 $O/r5q_image_ab.err
