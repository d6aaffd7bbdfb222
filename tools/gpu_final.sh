#!/bin/bash
# Final-tree measurement session of a round (after the last kernel-source change): `gpurun --timeout 3300 -- 'bash tools/gpu_final.sh r6'`.
# = tools/gpu_session.sh <tag> pytest smoke oproof stats:fp32 pmc:fp32:<dominant kernel> bench  (run first, separately, when the GPU-minute budget is tight)
# + the secondary figures the docs quote: the 8-tile image a rank sees at N = 8, BASELINE configs[1] (ViT-S), rocprofv3 stats of whole image passes,
#   the attention roofline in a fresh process.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp; O=gpurun_out; TAG=${1:-r6}
( timeout 900 python bench.py --steps 20 --warmup 5 ) > $O/${TAG}_bench_default.log 2> $O/${TAG}_bench_default.err
( timeout 600 python bench.py --split 2x4 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary ) > $O/${TAG}_shard_times.log 2>&1
( timeout 600 python bench.py --encoder vits --process-num 4 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline ) > $O/${TAG}_bench_vits.log 2>&1
( timeout 300 python -c "
import json, torch, bench
print(json.dumps(bench.roofline_attention(torch.device('cuda', 0))))" ) > $O/${TAG}_attention_fresh_process.log 2>&1
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/${TAG}_image_stats -o im -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary ) > $O/${TAG}_image_stats.log 2>&1
python tools/rocprof_summary.py "$(find $O/${TAG}_image_stats -name '*kernel_trace.csv' | head -1)" $O/${TAG}_image_kernel_stats.md "one f32 image pass x3 (1 warm-up + 2 timed), final tree" >> $O/${TAG}_image_stats.log 2>&1
find $O/${TAG}_image_stats -name '*kernel_trace.csv' -delete
tail -n 3 $O/${TAG}_bench_default.log | cut -c1-400; tail -n 2 $O/${TAG}_shard_times.log | cut -c1-300; tail -n 2 $O/${TAG}_bench_vits.log | cut -c1-300; cat $O/${TAG}_attention_fresh_process.log | tail -n 2 | cut -c1-400; head -20 $O/${TAG}_image_kernel_stats.md
