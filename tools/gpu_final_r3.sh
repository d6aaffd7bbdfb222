#!/bin/bash
# round-3 final verification + measurements (everything that ends up under profiles/ as r3_* "final tree" files)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
# 1. PMC passes on the dominant launch of each mode (separate --pmc runs, kernel-trace only); the summaries go to profiles/ ON THE BOX first so
#    that the bench line below can report roofline.traffic for exactly this kernel source
for dt in fp32; do
  P="python bench.py --roofline-only --dtype $dt"
  ( timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r3f_pmc_${dt}_a -o p -- $P ) > $O/r3f_pmc_${dt}_a.log 2>&1
  ( timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r3f_pmc_${dt}_b -o p -- $P ) > $O/r3f_pmc_${dt}_b.log 2>&1
  ( timeout 200 rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/r3f_pmc_${dt}_c -o p -- $P ) > $O/r3f_pmc_${dt}_c.log 2>&1
done
python tools/pmc_summary.py fp32 gemm_split3_kernel $O/r3_pmc_dominant_fp32.json $O/r3f_pmc_fp32_a $O/r3f_pmc_fp32_b $O/r3f_pmc_fp32_c > $O/r3f_pmc_fp32_summary.log 2>&1
cp $O/r3_pmc_dominant_fp32.json profiles/ 2>/dev/null
rm -f $O/r3f_pmc_*/p_kernel_trace.csv
# 2. the whole GPU suite, smoke, then the bench line exactly as the driver runs it
( timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16 ) > $O/r3_pytest_gpu.log 2>&1
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > $O/r3f_smoke.log 2>&1
( timeout 600 python bench.py ) > $O/r3_bench_default.json 2> $O/r3f_bench.err
# 3. rocprofv3 --stats of the roofline command and of a whole pass
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r3f_ro_f32 -o ro -- python bench.py --roofline-only --dtype fp32 ) > $O/r3_roofline_only_fp32.json 2> $O/r3f_ro_f32.log
# 4. BASELINE configs[1]
( timeout 300 python bench.py --encoder vits --process-num 4 --steps 10 --warmup 2 --no-cpu-baseline --no-roofline ) > $O/r3_bench_vits.json 2> $O/r3f_bench_vits.err
# 5. per-(op, shape) table of the f32 pass; all-16-tile headline parity
( timeout 400 python tools/op_roofline.py fp32 $O/r3_op_roofline_fp32.md $O/r3_op_roofline_fp32.json ) > $O/r3f_op_roofline.log 2>&1
( PF_HEADLINE_ALL=1 timeout 600 python -m pytest tests/test_headline_parity_gpu.py -m gpu -x -q -k "fp32" 2>&1 | tail -3 ) > $O/r3f_headline_all.log 2>&1
rm -f $O/r3f_*/*kernel_trace.csv $O/r3f_*/*/*kernel_trace.csv
cat $O/r3_pytest_gpu.log; cat $O/r3f_smoke.log; cut -c1-700 $O/r3_bench_default.json; cat $O/r3f_pmc_fp32_summary.log | tail -30; cut -c1-400 $O/r3_bench_vits.json; cat $O/r3f_headline_all.log
