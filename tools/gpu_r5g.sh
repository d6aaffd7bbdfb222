#!/bin/bash
# round 5, call 7: 16-byte paired plane stores of the Winograd input transform (PF_W3_PAIR); per-kernel totals of an image pass (rocprofv3)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 900 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "conv_winograd" 2>&1 | tail -4 ) > $O/r5g_checks.log 2>&1
echo "== checks"; cat $O/r5g_checks.log
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "PF_W3_PAIR=0" "PF_W3_PAIR=1" ) > $O/r5g_image_ab.md 2> $O/r5g_image_ab.err
echo "== image ab"; cat $O/r5g_image_ab.md; tail -2 $O/r5g_image_ab.err
for pr in 0 1; do
  rm -rf /tmp/rp$pr
  ( PF_W3_PAIR=$pr timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rp$pr -o ro -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary ) > $O/r5g_stats_pair$pr.log 2>&1
  f=$(find /tmp/rp$pr -name '*kernel_trace.csv' | head -1)
  python tools/rocprof_summary.py "$f" $O/r5g_image_kernel_stats_pair$pr.md "one f32 image pass x3 (1 warm-up + 2 timed), PF_W3_PAIR=$pr" > /dev/null 2>&1
  echo "== stats pair=$pr"; head -30 $O/r5g_image_kernel_stats_pair$pr.md
done
