"""Summarise rocprofv3 --pmc passes of ONE kernel (the dominant launch of `bench.py --roofline-only`) into a JSON for profiles/.

usage: python tools/pmc_summary.py <dtype> <kernel-substring> <out.json> <pass_dir> [<pass_dir> ...]
Every pass dir holds *_counter_collection.csv of one `rocprofv3 --pmc ... --kernel-trace --output-format csv` run of
`bench.py --roofline-only`, which calls pf_conv CALLS = 6 times (1 warm-up + 5 timed).  One pf_conv call may be several
dispatches (the f32 channel split: body + remainder), so every counter is SUMMED over all matching dispatches of a pass and
divided by CALLS -> per pf_conv launch, the same unit as roofline.achieved.
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs;
FETCH_SIZE (KB) reports 1/2 of a wide coalesced read stream on gfx950 -> doubled; WRITE_SIZE (KB) as is.
"""
import csv
import glob
import hashlib
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    dtype, sub, out = sys.argv[1], sys.argv[2], sys.argv[3]
    vals, durs, grid = defaultdict(list), [], 0
    rows = []
    for d in sys.argv[4:]:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if sub in r["Kernel_Name"]:
                        r["__pass"] = d
                        rows.append(r)
    if not rows:
        print("no rows for", sub)
        return
    CALLS = int(os.environ.get("PF_PMC_CALLS", "6"))      # pf_conv / pf_gemm_split3 calls of one `bench.py --roofline-only` run (the bench prints "calls": n)
    for d in sys.argv[4:]:                                 # ... or read it from the run's own JSON line (log beside the pass directory)
        try:
            for ln in open(d + ".log"):
                if ln.startswith("{") and '"calls"' in ln:
                    CALLS = int(json.loads(ln)["calls"])
        except Exception:
            pass
    name = sorted({r["Kernel_Name"] for r in rows}, key=len)[0]
    seen = defaultdict(set)
    tot = defaultdict(float)
    dur_by_pass = defaultdict(float)
    for r in rows:
        key = (r["__pass"], r["Dispatch_Id"])
        tot[r["Counter_Name"]] += float(r["Counter_Value"])
        if key not in seen["d"]:
            seen["d"].add(key)
            dur_by_pass[r["__pass"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    # a counter lives in exactly one pass -> its sum over that pass / CALLS
    per = {k: v / CALLS for k, v in tot.items()}
    ms = sum(dur_by_pass.values()) / len(dur_by_pass) / CALLS
    durs = list(seen["d"])
    grid = sorted({int(r["Grid_Size"]) for r in rows})
    der = {"ms_per_launch_profiled": ms}
    if "GRBM_GUI_ACTIVE" in per:
        # GRBM_GUI_ACTIVE may be collected in several passes: use the mean per pass
        n_grbm = sum(1 for d in dur_by_pass if any(r["__pass"] == d and r["Counter_Name"] == "GRBM_GUI_ACTIVE" for r in rows))
        cyc = per["GRBM_GUI_ACTIVE"] / max(n_grbm, 1) / 8.0
        der["kernel_cycles_per_xcd"] = cyc
        der["effective_clock_GHz"] = cyc / (ms * 1e-3) / 1e9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in per:
            der["mfma_pipe_busy_frac"] = per["SQ_VALU_MFMA_BUSY_CYCLES"] / (256 * 4) / cyc
    if "SQ_WAVE_CYCLES" in per:
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_SCA"):
            if k in per:
                der[k + "_frac_of_wave_cycles"] = per[k] / per["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in per:
        der["hbm_read_bytes_corrected_x2"] = per["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in per:
        der["hbm_write_bytes"] = per["WRITE_SIZE"] * 1024
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        der["traffic_bytes"] = der["hbm_read_bytes_corrected_x2"] + der["hbm_write_bytes"]
    h = hashlib.sha256()
    for f in ("igemm.hip", "wino_fused.hip", "gemm_split3.hip", "winograd.hip", "pf_common.h"):         # == bench.py kernel_source_sha()
        h.update(open(os.path.join(ROOT, "patchfusion_amd", "csrc", f), "rb").read())
    wm = 0
    if dtype == "fp32":
        sys.path.insert(0, ROOT)
        from patchfusion_amd.packing import winograd_mode
        wm = winograd_mode()
    j = {"kernel": name, "grid": grid, "dtype": dtype, "kernel_source_sha": h.hexdigest()[:12], "winograd_m": wm, "ws_cap_gb": os.environ.get("PF_WS_CAP_GB", "2.5"), "dispatches_total_all_passes": len(durs), "pf_conv_calls_per_pass": CALLS,
         "command": "rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python bench.py --roofline-only --dtype " + dtype,
         "per_launch": per, "derived": der}
    json.dump(j, open(out, "w"), indent=1)
    print(json.dumps(der, indent=1))


if __name__ == "__main__":
    main()
