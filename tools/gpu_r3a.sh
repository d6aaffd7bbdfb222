#!/bin/bash
# round-3 GPU session A: fused Winograd kernel bring-up (correctness -> timing sweep), new parity tests, e2e A/B, configs[1] (vits) bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 400 python tools/wino_fused_probe.py check ) > $O/r3a_wf_check.log 2>&1; echo "check rc=$?" >> $O/r3a_wf_check.log
tail -16 $O/r3a_wf_check.log
if grep -q "^0 failing cases" $O/r3a_wf_check.log; then
  ( timeout 500 python tools/wino_fused_probe.py time ) > $O/r3a_wf_time.log 2>&1
  cat $O/r3a_wf_time.log
  ( timeout 600 python -m pytest tests/test_hip_ops_gpu.py -q -x -k "winograd" 2>&1 | tail -5 ) > $O/r3a_pytest_wino.log 2>&1
  cat $O/r3a_pytest_wino.log
fi
( timeout 900 python -m pytest tests/test_e2e_gpu.py -q -x -k "configs3 or rccl" -s 2>&1 | tail -12 ) > $O/r3a_pytest_new.log 2>&1
cat $O/r3a_pytest_new.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
( PF_WINO_FUSED=0 PF_VIT_BATCH_ALL=0 timeout 300 $B ) > $O/r3a_bench_base.json 2> $O/r3a_bench_base.err
( PF_WINO_FUSED=0 PF_VIT_BATCH_ALL=1 timeout 300 $B ) > $O/r3a_bench_vitall.json 2> $O/r3a_bench_vitall.err
if grep -q "^0 failing cases" $O/r3a_wf_check.log; then
  ( PF_WINO_FUSED=1 PF_VIT_BATCH_ALL=1 timeout 300 $B ) > $O/r3a_bench_fused_vitall.json 2> $O/r3a_bench_fused_vitall.err
fi
for f in base vitall fused_vitall; do echo "== $f"; cat $O/r3a_bench_$f.json 2>/dev/null | head -c 600; echo; tail -2 $O/r3a_bench_$f.err 2>/dev/null; done
# BASELINE configs[1]: DA-vits, 4K, 4x4 m1, process_num 4
( timeout 300 python bench.py --encoder vits --process-num 4 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline ) > $O/r3a_bench_vits.json 2> $O/r3a_bench_vits.err
cat $O/r3a_bench_vits.json | head -c 1500; echo
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3a_prof_vits -o vits -- python bench.py --encoder vits --process-num 4 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary ) > $O/r3a_prof_vits.log 2>&1
ls $O/r3a_prof_vits/ 2>/dev/null | head; find $O/r3a_prof_vits -name "*kernel_stats.csv" | head -1 | xargs -r head -25
find $O/r3a_prof_vits -name "*kernel_trace.csv" -size +20M -delete 2>/dev/null
