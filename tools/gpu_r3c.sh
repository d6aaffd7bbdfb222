#!/bin/bash
# round-3 GPU session C: where does the fused Winograd kernel's time go?  debug-switch decomposition + PMC passes
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python tools/wino_fused_probe.py decomp c544_544,c544_32,c768_768_L4 ) > $O/r3c_decomp.log 2>&1
cat $O/r3c_decomp.log
P="python tools/wino_fused_probe.py one c544_544"
( timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r3c_pmc_a -o p -- $P ) > $O/r3c_pmc_a.log 2>&1
( timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r3c_pmc_b -o p -- $P ) > $O/r3c_pmc_b.log 2>&1
( timeout 200 rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/r3c_pmc_c -o p -- $P ) > $O/r3c_pmc_c.log 2>&1
( timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $O/r3c_pmc_d -o p -- $P ) > $O/r3c_pmc_d.log 2>&1
rm -f $O/r3c_pmc_*/*/*kernel_trace.csv $O/r3c_pmc_*/*kernel_trace.csv
python - <<'PY'
import csv, glob, collections
for d in "abcd":
    tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
    for f in glob.glob(f"gpurun_out/r3c_pmc_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino_fused" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k in tot:
        print(f"pass {d}: {k} = {tot[k] / max(n[k], 1):.4g} per launch ({n[k]} launches, {sum(dur) / max(len(dur), 1):.3f} ms avg)")
PY
