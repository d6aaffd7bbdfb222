#!/bin/bash
mkdir -p gpurun_out
for sp in 2x4 4x4; do
  timeout 300 python bench.py --split $sp --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('split $sp', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3an_shard_times.log
done
