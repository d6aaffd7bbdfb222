#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
( timeout 100 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r3ap_vits -o vits -- python bench.py --encoder vits --process-num 4 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-roofline ) > gpurun_out/r3ap_vits.log 2>&1
tail -1 gpurun_out/r3ap_vits.log | cut -c1-200
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r3ap_vits/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
starts = [s for s, e, n in rows if "patch_im2col" in n]
# the timed passes = after the first im2col of the 2nd pass ... end; take the last 3 passes: find pass boundaries by crop_resize kernels
first = [s for s, e, n in rows if "crop_resize_planar" in n]
t0 = first[-6] if len(first) >= 6 else first[0]        # two crop launches per pass (two batches) -> last three passes
sel = [(s, e, n) for s, e, n in rows if s >= t0]
t1 = max(e for s, e, n in sel)
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, n in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
small = sum(e - s for s, e, n in sel if e - s < 10000)
tot = sum(e - s for s, e, n in sel)
print(f"vits last passes: wall {(t1 - t0) / 1e6:.1f} ms, GPU busy (union of kernel intervals) {busy / 1e6:.1f} ms = {100 * busy / (t1 - t0):.1f} %, "
      f"{len(sel)} kernels, kernels < 10 us: {100 * small / tot:.1f} % of kernel time, sum of kernel times {tot / 1e6:.1f} ms")
PY
rm -rf gpurun_out/r3ap_vits
