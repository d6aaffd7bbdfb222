#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out
( time timeout 900 python bench.py ) > gpurun_out/r5t_bench_noflags.json 2> gpurun_out/r5t_bench_noflags.err
python - <<'PY'
import json
j=json.loads([l for l in open('gpurun_out/r5t_bench_noflags.json') if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], j['memory'], j['roofline']['frac'], j['roofline']['traffic'], j['bf16']['roofline']['traffic'], j['cpu_baseline']['value'])
PY
tail -4 gpurun_out/r5t_bench_noflags.err
