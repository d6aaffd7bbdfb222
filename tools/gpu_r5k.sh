#!/bin/bash
# round 5, call 11: workspace cap sweep, bench line sanity
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "PF_WS_CAP_GB=100" "PF_WS_CAP_GB=5" "PF_WS_CAP_GB=2.5" "PF_WS_CAP_GB=1.3" "PF_WS_CAP_GB=0.7" ) > $O/r5k_image_ab.md 2> $O/r5k_image_ab.err
echo "== image ab"; cat $O/r5k_image_ab.md; tail -2 $O/r5k_image_ab.err
( timeout 600 python bench.py --steps 5 --warmup 2 --no-secondary --no-cpu-baseline ) > $O/r5k_bench.json 2> $O/r5k_bench.err
echo "== bench"; cat $O/r5k_bench.json; tail -3 $O/r5k_bench.err
