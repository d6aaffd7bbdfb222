#!/bin/bash
# round-2 GPU call 2: new op checks + headline parity + f32 diagnosis (PMC, kernel trace) + per-op roofline tables
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python -m pytest tests/test_hip_ops_gpu.py -m gpu -q -x -k "split or dominant or vitl_linear or n544 or n160 or conv_gemm" 2>&1 | tail -5 ) > $O/r2c2_opchecks.log 2>&1
( timeout 400 python -m pytest tests/test_headline_parity_gpu.py -m gpu -q -x 2>&1 | tail -8 ) > $O/r2c2_headline.log 2>&1
( timeout 300 python -m pytest tests/test_e2e_gpu.py -m gpu -q -s -k "tiny_bf16 or vitl_patch_batch" 2>&1 | grep -E "MEASURED|passed|failed" ) > $O/r2c2_e2e_measured.log 2>&1
( timeout 200 python bench.py --gemm-sweep --dtype fp32 ) > $O/r2c2_sweep_fp32.log 2>&1
for c in 1 2 3 6; do ( PF_IGEMM_CFG=$c timeout 100 python bench.py --gemm-sweep --dtype fp32 --only qkv,proj,fc1,fc2,up1_768_L1,c512_256_L1,c512_256_L0 ) > $O/r2c2_sweep_fp32_cfg$c.log 2>&1; done
# PMC on the f32 dominant launch (unsplit = the kernel itself), separate passes
P="python bench.py --roofline-only --dtype fp32"
( PF_IGEMM_NOSPLIT=1 timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_f32_a -o p -- $P ) > $O/pmc_f32_a.log 2>&1
( PF_IGEMM_NOSPLIT=1 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f32_b -o p -- $P ) > $O/pmc_f32_b.log 2>&1
( PF_IGEMM_NOSPLIT=1 timeout 200 rocprofv3 --pmc WRITE_SIZE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_f32_c -o p -- $P ) > $O/pmc_f32_c.log 2>&1
python tools/pmc_summary.py fp32 conv_igemm_kernel $O/r2_pmc_dominant_fp32.json $O/pmc_f32_a $O/pmc_f32_b $O/pmc_f32_c > $O/pmc_f32_summary.log 2>&1
# kernel trace of one f32 image pass
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -o bench -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline ) > $O/prof_f32.log 2>&1
# per-op roofline tables
( timeout 300 python tools/op_roofline.py fp32 $O/r2_op_roofline_fp32.md $O/r2_op_roofline_fp32.json ) > $O/op_roofline_fp32.log 2>&1
( timeout 300 python tools/op_roofline.py bf16 $O/r2_op_roofline_bf16.md $O/r2_op_roofline_bf16.json ) > $O/op_roofline_bf16.log 2>&1
rm -f $O/pmc_f32_*/p_kernel_trace.csv $O/prof_f32/bench_kernel_trace.csv.bak
tail -3 $O/r2c2_opchecks.log; tail -6 $O/r2c2_headline.log; cat $O/r2c2_e2e_measured.log; cat $O/r2c2_sweep_fp32.log; tail -8 $O/r2c2_sweep_fp32_cfg*.log; cat $O/pmc_f32_summary.log; head -40 $O/op_roofline_fp32.log
