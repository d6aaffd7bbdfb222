#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
P="python tools/attn_probe.py fp32 10"
( timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r3m_pmc_a -o p -- $P ) > $O/r3m_pmc_a.log 2>&1
( timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/r3m_pmc_c -o p -- $P ) > $O/r3m_pmc_c.log 2>&1
rm -f $O/r3m_pmc_*/*kernel_trace.csv
python - <<'PY'
import csv, glob, collections
for d in "ac":
    tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
    for f in glob.glob(f"gpurun_out/r3m_pmc_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "vit_attention_qkv" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in tot:
        print(f"pass {d}: {k} = {tot[k] / max(n[k], 1):.4g} per launch ({n[k]} launches, {sum(dur) / max(len(dur), 1):.1f} us avg)")
PY
