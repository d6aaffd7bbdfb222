#!/bin/bash
mkdir -p gpurun_out
for pp in 1 0; do
  PF_S3_PP=$pp timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PF_S3_PP=$pp', d['ms_per_step'], d['value'])" | tee -a gpurun_out/r3af_bench.log
done
