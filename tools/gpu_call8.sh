#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python tools/f32_tune.py $O/r2_f32_tune_korder.json ) > $O/r2_f32_tune_korder.log 2>&1
( timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "fp32 and conv and not dominant" 2>&1 | tail -4 ) > $O/r2c8_opchecks.log 2>&1
P="python bench.py --roofline-only --dtype fp32"
( timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f32k_b -o p -- $P ) > $O/pmc_f32k_b.log 2>&1
( timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_f32k_a -o p -- $P ) > $O/pmc_f32k_a.log 2>&1
( timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_f32k_c -o p -- $P ) > $O/pmc_f32k_c.log 2>&1
python tools/pmc_summary.py fp32 conv_igemm_kernel $O/r2_pmc_dominant_fp32_korder.json $O/pmc_f32k_a $O/pmc_f32k_b $O/pmc_f32k_c > $O/pmc_f32k_summary.log 2>&1
( timeout 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary ) > $O/r2c8_bench.json 2> $O/r2c8_bench.err
rm -f $O/pmc_f32k_*/p_kernel_trace.csv
cat $O/r2_f32_tune_korder.log; tail -n 3 $O/r2c8_opchecks.log; cat $O/pmc_f32k_summary.log; cat $O/r2c8_bench.json
