#!/bin/bash
# session U: PMC of the split GEMM (what bounds it)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
P="python tools/split3_probe.py 8296"
( timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/r3u_pmc_a -o p -- $P ) > $O/r3u_pmc_a.log 2>&1
( timeout 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O/r3u_pmc_b -o p -- $P ) > $O/r3u_pmc_b.log 2>&1
( timeout 200 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/r3u_pmc_c -o p -- $P ) > $O/r3u_pmc_c.log 2>&1
rm -f $O/r3u_pmc_*/*kernel_trace.csv $O/r3u_pmc_*/*/*kernel_trace.csv
tail -2 $O/r3u_pmc_b.log
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/r3u_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_split3" in r["Kernel_Name"]:
            key = (r["Kernel_Name"][-40:], r["Grid_Size"], r["Counter_Name"])
            tot[key] += float(r["Counter_Value"]); n[key] += 1
for k in sorted(tot): print(k, f"{tot[k]/n[k]:.5g}", n[k])
PY
