"""Run every per-op HIP parity check in its own subprocess (a faulting kernel cannot hide the rest) and
write a JSON + text report under gpurun_out/.  Usage on the GPU box:  python tools/gpu_selfcheck.py"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ONE = r"""
import sys, json, torch
sys.path.insert(0, %r)
from tests import op_checks
name, dt = sys.argv[1], sys.argv[2]
err, tol, info = op_checks.CHECKS[name](op_checks.DTYPES[dt])
print("RESULT " + json.dumps(dict(name=name, dtype=dt, err=err, tol=tol, info=info, ok=bool(err <= tol))))
""" % ROOT


def main():
    from tests import op_checks
    only = sys.argv[1:]
    out = []
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for name in op_checks.CHECKS:
        if only and name not in only:
            continue
        for dt in ("fp32", "bf16"):
            if dt == "bf16" and name in op_checks.F32_ONLY:
                continue
            t = time.time()
            try:
                r = subprocess.run([sys.executable, "-c", ONE, name, dt], capture_output=True, text=True, timeout=240, cwd=ROOT)
                line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
                if line:
                    res = json.loads(line[-1][7:])
                else:
                    res = dict(name=name, dtype=dt, ok=False, err=None, crash=(r.stderr or r.stdout)[-600:], rc=r.returncode)
            except subprocess.TimeoutExpired:
                res = dict(name=name, dtype=dt, ok=False, err=None, crash="TIMEOUT")
            res["secs"] = round(time.time() - t, 1)
            out.append(res)
            print(("PASS " if res["ok"] else "FAIL ") + json.dumps(res), flush=True)
    with open(os.path.join(ROOT, "gpurun_out", "selfcheck.json"), "w") as f:
        json.dump(out, f, indent=1)
    bad = [r for r in out if not r["ok"]]
    print(f"{len(out) - len(bad)}/{len(out)} op checks passed")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
