#!/bin/bash
# rocprofv3 counters of ONE kernel: three separate --pmc passes (--kernel-trace only, as MI355X_MICROARCH.md prescribes) + a --kernel-trace --stats pass over a
# command that launches the kernel N times, summarised per launch by tools/pmc_summary.py.
#   usage (GPU box): bash tools/kernel_pmc.sh <tag> <kernel-name-substring> <N> <python command ...>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; TAG=$1; KERN=$2; N=$3; shift 3; P="$*"
( timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_pmc_a -o p -- $P ) > $O/${TAG}_pmc_a.log 2>&1
( timeout 300 rocprofv3 --pmc FETCH_SIZE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/${TAG}_pmc_b -o p -- $P ) > $O/${TAG}_pmc_b.log 2>&1
( timeout 300 rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/${TAG}_pmc_c -o p -- $P ) > $O/${TAG}_pmc_c.log 2>&1
PF_PMC_CALLS=$N python tools/pmc_summary.py fp32 "$KERN" $O/${TAG}_pmc.json $O/${TAG}_pmc_a $O/${TAG}_pmc_b $O/${TAG}_pmc_c > $O/${TAG}_pmc_summary.log 2>&1
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o ro -- $P ) > $O/${TAG}_stats.log 2>&1
find $O/${TAG}_pmc_a $O/${TAG}_pmc_b $O/${TAG}_pmc_c $O/${TAG}_stats -name '*kernel_trace.csv' -delete
tail -n 22 $O/${TAG}_pmc_summary.log; find $O/${TAG}_stats -name '*kernel_stats.csv' | head -1 | xargs head -4 | cut -c1-220
