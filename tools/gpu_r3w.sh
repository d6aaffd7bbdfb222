#!/bin/bash
# session W: single-batch shard split A/B, all-16-tile headline parity (fixture), full default bench
mkdir -p gpurun_out
for m in 1 0; do
  PF_SPLIT_SINGLE_BATCH=$m timeout 300 python bench.py --split 2x4 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split_single_batch=$m', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3w_split_single.log
done
PF_HEADLINE_ALL=1 timeout 900 python -m pytest tests/test_headline_parity_gpu.py -m gpu -x -q -k "fp32 or fixture" > gpurun_out/r3w_headline_all.log 2>&1
tail -3 gpurun_out/r3w_headline_all.log
timeout 900 python bench.py > gpurun_out/r3w_bench.json 2> gpurun_out/r3w_bench.err
tail -1 gpurun_out/r3w_bench.json | cut -c1-1500
