#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary"
( PF_WINO_FUSED=1 timeout 300 $B ) > $O/r3k_bench_auto.json 2> $O/r3k_bench_auto.err
( PF_WINO_FUSED=2 timeout 300 $B --no-roofline ) > $O/r3k_bench_all.json 2> $O/r3k_bench_all.err
( PF_WINO_FUSED=0 timeout 300 $B --no-roofline ) > $O/r3k_bench_off.json 2> $O/r3k_bench_off.err
for f in auto all off; do echo "== $f"; python -c "import json,sys; j=json.load(open('$O/r3k_bench_$f.json')); print(j['ms_per_step'], j.get('roofline',{}).get('frac'), j.get('roofline',{}).get('ms_per_launch'))"; done
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -14 ) > $O/r3k_pytest_gpu.log 2>&1
cat $O/r3k_pytest_gpu.log
