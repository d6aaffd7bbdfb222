#!/bin/bash
# round 5, call 1: MFMA ceiling microbenchmark, transform/GEMM overlap probe, image-pass A/B of grid caps and tile order
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 120 tools/mfma_ceiling 40 ) > $O/r5a_mfma_ceiling.md 2>&1
echo "== mfma"; cat $O/r5a_mfma_ceiling.md
( timeout 600 python tools/overlap_probe.py 6 ) > $O/r5a_overlap_probe.md 2>&1
echo "== overlap"; tail -12 $O/r5a_overlap_probe.md
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "" "PF_S3_GRID=224" "PF_S3_GRID=192" "PF_S3_GRID=128" "PF_S3_ORDER=2" "PF_W3_GRID=192" "PF_W3_GRID=128" ) > $O/r5a_image_ab.md 2> $O/r5a_image_ab.err
echo "== image ab"; cat $O/r5a_image_ab.md; tail -3 $O/r5a_image_ab.err
