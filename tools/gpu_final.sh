#!/bin/bash
# round-2 final verification + measurements (everything that ends up under profiles/ as r2c_* / r2_pmc_dominant_*)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
# 1. PMC passes on the dominant launch of each mode (separate --pmc runs, kernel-trace only); the summaries go to profiles/ ON THE BOX
#    first so that the bench line below can report roofline.traffic for exactly this kernel source
for dt in fp32 bf16; do
  P="python bench.py --roofline-only --dtype $dt"
  ( timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmcf_${dt}_a -o p -- $P ) > $O/pmcf_${dt}_a.log 2>&1
  ( timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_${dt}_b -o p -- $P ) > $O/pmcf_${dt}_b.log 2>&1
  ( timeout 200 rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $O/pmcf_${dt}_c -o p -- $P ) > $O/pmcf_${dt}_c.log 2>&1
done
python tools/pmc_summary.py fp32 conv_igemm_kernel $O/r2_pmc_dominant_fp32.json $O/pmcf_fp32_a $O/pmcf_fp32_b $O/pmcf_fp32_c > $O/pmcf_fp32_summary.log 2>&1
python tools/pmc_summary.py bf16 conv3x3_halo_kernel $O/r2_pmc_dominant_bf16.json $O/pmcf_bf16_a $O/pmcf_bf16_b $O/pmcf_bf16_c > $O/pmcf_bf16_summary.log 2>&1
cp $O/r2_pmc_dominant_fp32.json $O/r2_pmc_dominant_bf16.json profiles/ 2>/dev/null
rm -f $O/pmcf_*/p_kernel_trace.csv
# 2. the whole GPU suite, then the bench line exactly as the driver runs it
( timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16 ) > $O/r2f_pytest_gpu.log 2>&1
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 ) > $O/r2f_smoke.log 2>&1     # build() + smoke() in ONE process, like a driver may run them
( timeout 300 python bench.py --steps 5 --warmup 2 ) > $O/r2f_bench.json 2> $O/r2f_bench.err
# 3. rocprofv3 --stats of the roofline command (average duration of the dominant kernel next to the HIP-event figure) and of a whole pass
( timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proff_ro_f32 -o ro -- python bench.py --roofline-only --dtype fp32 ) > $O/r2_roofline_only_fp32.json 2> $O/proff_ro_f32.log
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proff_f32 -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline ) > $O/proff_f32.log 2>&1
# 4. per-(op, shape) table of the f32 pass
( timeout 300 python tools/op_roofline.py fp32 $O/r2_op_roofline_fp32.md $O/r2_op_roofline_fp32.json ) > $O/op_roofline_fp32.log 2>&1
# 5. A/B of the Winograd tile size: F(2x2,3x3) (the direct kernels, PF_WINOGRAD=0, were measured in r2b_bench_direct_f32.json)
( PF_WINOGRAD=2 timeout 200 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary ) > $O/r2f_bench_wino2.json 2> $O/r2f_bench_wino2.err
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/proff_bf16 -o bench -- python bench.py --dtype bf16 --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline ) > $O/proff_bf16.log 2>&1
cat $O/r2f_pytest_gpu.log; cat $O/r2f_bench.json; cat $O/pmcf_fp32_summary.log $O/pmcf_bf16_summary.log; cat $O/r2f_bench_wino2.json; sed -n 1,24p $O/r2_op_roofline_fp32.md
