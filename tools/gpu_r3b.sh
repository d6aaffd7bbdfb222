#!/bin/bash
# round-3 GPU session B: fused Winograd v2 (2-D super-tiles, 16-channel swizzled raw stage): correctness -> timing -> e2e
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 400 python tools/wino_fused_probe.py check ) > $O/r3b_wf_check.log 2>&1; echo "check rc=$?" >> $O/r3b_wf_check.log
tail -17 $O/r3b_wf_check.log
if grep -q "^0 failing cases" $O/r3b_wf_check.log; then
  ( timeout 500 python tools/wino_fused_probe.py time ) > $O/r3b_wf_time.log 2>&1
  cat $O/r3b_wf_time.log
  ( timeout 600 python -m pytest tests/test_hip_ops_gpu.py -q -x -k "winograd" 2>&1 | tail -5 ) > $O/r3b_pytest_wino.log 2>&1
  cat $O/r3b_pytest_wino.log
fi
( timeout 900 python -m pytest tests/test_e2e_gpu.py -q -k "configs3 or rccl" -s 2>&1 | tail -12 ) > $O/r3b_pytest_new.log 2>&1
cat $O/r3b_pytest_new.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
if grep -q "^0 failing cases" $O/r3b_wf_check.log; then
  ( PF_WINO_FUSED=1 timeout 300 $B ) > $O/r3b_bench_fused.json 2> $O/r3b_bench_fused.err
  echo "== fused"; head -c 400 $O/r3b_bench_fused.json; echo; tail -2 $O/r3b_bench_fused.err
fi
