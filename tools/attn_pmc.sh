#!/bin/bash
# rocprofv3 PMC passes (separate passes, --kernel-trace only) + a --stats pass over N launches of the version-2 split attention (tools/attn_v2_probe.py launch),
# summarised by tools/pmc_summary.py.   usage (GPU box): bash tools/attn_pmc.sh <tag> [qw] [B]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out; TAG=$1; QW=${2:-32}; BB=${3:-8}; N=12
P="python tools/attn_v2_probe.py launch $N $QW $BB"
( timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${TAG}_pmc_a -o p -- $P ) > $O/${TAG}_pmc_a.log 2>&1
( timeout 300 rocprofv3 --pmc FETCH_SIZE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/${TAG}_pmc_b -o p -- $P ) > $O/${TAG}_pmc_b.log 2>&1
( timeout 300 rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/${TAG}_pmc_c -o p -- $P ) > $O/${TAG}_pmc_c.log 2>&1
PF_PMC_CALLS=$N python tools/pmc_summary.py fp32 vit_attention_split3_ $O/${TAG}_pmc_attention_v2.json $O/${TAG}_pmc_a $O/${TAG}_pmc_b $O/${TAG}_pmc_c
( timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats_attn -o ro -- $P ) > $O/${TAG}_stats_attn.log 2>&1
find $O -name '*kernel_trace.csv' -delete
cat $O/${TAG}_stats_attn/*/ro_kernel_stats.csv 2>/dev/null | head -5 || find $O/${TAG}_stats_attn -name '*stats*' | head
