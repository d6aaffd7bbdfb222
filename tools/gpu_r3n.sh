#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 300 python -m pytest tests/test_hip_ops_gpu.py -q -k "vit_attention" 2>&1 | tail -3 ) > $O/r3n_pytest_attn.log 2>&1; cat $O/r3n_pytest_attn.log
( PF_ATTN_QKV=1 timeout 100 python tools/attn_probe.py fp32 20 ) 2>&1 | grep vit_attention > $O/r3n_attn_time.log; cat $O/r3n_attn_time.log
P="python tools/attn_probe.py fp32 10"
( timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/r3n_pmc -o p -- $P ) > $O/r3n_pmc.log 2>&1
rm -f $O/r3n_pmc/*kernel_trace.csv
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob("gpurun_out/r3n_pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "vit_attention_qkv" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in tot: print(k, f"{tot[k]/n[k]:.4g}")
PY
