#!/bin/bash
# round-3 GPU session D: fused Winograd v3 (bank-rotated V image, 12-op transforms): correctness -> timing -> decomposition -> PMC -> e2e
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 400 python tools/wino_fused_probe.py check ) > $O/r3d_wf_check.log 2>&1; echo "check rc=$?" >> $O/r3d_wf_check.log
tail -4 $O/r3d_wf_check.log
if grep -q "^0 failing cases" $O/r3d_wf_check.log; then
  ( timeout 500 python tools/wino_fused_probe.py time ) > $O/r3d_wf_time.log 2>&1
  cut -c1-60,250-420 $O/r3d_wf_time.log
  ( PF_LIB_PATH=$PWD/patchfusion_amd/libpf_abl_wfdbg.so timeout 300 python tools/wino_fused_probe.py decomp c544_544,c544_32 ) > $O/r3d_decomp.log 2>&1
  cat $O/r3d_decomp.log
  P="python tools/wino_fused_probe.py one c544_544"
  ( timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r3d_pmc_a -o p -- $P ) > $O/r3d_pmc_a.log 2>&1
  ( timeout 200 rocprofv3 --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/r3d_pmc_c -o p -- $P ) > $O/r3d_pmc_c.log 2>&1
  rm -f $O/r3d_pmc_*/*kernel_trace.csv
  python - <<'PY'
import csv, glob, collections
for d in "ac":
    tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
    for f in glob.glob(f"gpurun_out/r3d_pmc_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wino_fused" in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    for k in tot:
        print(f"pass {d}: {k} = {tot[k] / max(n[k], 1):.4g} per launch ({n[k]} launches, {sum(dur) / max(len(dur), 1):.3f} ms avg)")
PY
  B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-roofline"
  ( PF_WINO_FUSED=1 timeout 300 $B ) > $O/r3d_bench_fused.json 2> $O/r3d_bench_fused.err
  echo "== fused"; head -c 300 $O/r3d_bench_fused.json; echo; tail -1 $O/r3d_bench_fused.err
fi
