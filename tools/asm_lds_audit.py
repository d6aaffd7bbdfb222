"""Audit of hand-counted LDS reads in a gfx950 assembly listing (hipcc -save-temps).

The tap-reuse conv kernels issue `ds_read_b128` from inline asm and wait with hand-counted
`s_waitcnt lgkmcnt(N)`: hipcc does not know those registers are in flight, so a register copy, spill or
early use it schedules between a read and the wait that releases it would silently read stale data.
LDS returns in order, so the check is a queue simulation per kernel: every `ds_read*` pushes its
destination registers; `s_waitcnt lgkmcnt(N)` retires all but the youngest N; any other instruction that
touches a register still in flight is a violation.  Loops are handled by walking the listing linearly twice
per kernel body (a back edge only ever finds an empty or identical queue in these kernels because every
step ends with lgkmcnt(0) before its barrier).

usage: python tools/asm_lds_audit.py file.s [kernel-name-substring]   -> exit status 1 on violations
"""
import re
import sys


def _regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def audit(path, name_filter="conv3x3_halo_kernel"):
    text = open(path).read()
    bad, checked = [], 0
    for m in re.finditer(r"^(_Z\w+):[^\n]*$", text, re.M):
        name = m.group(1)
        if name_filter not in name:
            continue
        end = text.index(".Lfunc_end", m.end())
        inflight = []          # list of (regset, line) in issue order
        nreads = 0
        for ln in text[m.end():end].split("\n"):
            s = ln.strip()
            if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
                continue
            s = s.split(";")[0].strip()
            if not s:
                continue
            op = s.split()[0]
            if op.startswith("ds_read"):
                dst = s[len(op):].split(",")[0]
                inflight.append((_regs(dst), s))
                nreads += 1
                # the address operand must not be in flight either
                rest = ",".join(s[len(op):].split(",")[1:])
                for rs, src in inflight[:-1]:
                    if rs & _regs(rest):
                        bad.append((name, s, src))
                continue
            if op == "s_waitcnt":
                mm = re.search(r"lgkmcnt\((\d+)\)", s)
                if mm:
                    n = int(mm.group(1))
                    inflight = inflight[len(inflight) - n:] if n else []
                continue
            if op in ("s_barrier",) or op.startswith("s_"):
                continue
            used = _regs(s[len(op):])
            for rs, src in inflight:
                if rs & used:
                    bad.append((name, s, src))
        checked += 1 if nreads else 0
    return checked, bad


if __name__ == "__main__":
    n, bad = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "conv3x3_halo_kernel")
    print(f"{n} kernels with LDS reads audited, {len(bad)} violations")
    for name, ins, src in bad[:20]:
        print(f"  {name[:60]}: `{ins}` touches registers of in-flight `{src}`")
    sys.exit(1 if bad else 0)
