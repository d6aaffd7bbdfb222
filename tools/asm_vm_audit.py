"""Audit of hand-counted VMEM loads in a gfx950 assembly listing (hipcc -save-temps).

csrc/wino_fused.hip issues its filter-fragment loads (`global_load_dwordx4 v[..]`) and the LDS-DMA (`global_load_lds_dwordx4`)
from inline asm and waits with hand-counted `s_waitcnt vmcnt(N)`.  hipcc does not know those registers are in flight: a copy, a
spill or an early use scheduled between a load and the wait that releases it would silently read stale data.  VMEM returns in
order (for vmcnt purposes), so the check is a queue simulation: every load pushes its destination registers (the DMA pushes an
entry without registers), `s_waitcnt vmcnt(N)` retires all but the youngest N, any other instruction touching a register that
is still in flight is a violation -- including a second load into it.

Control flow: (1) the whole kernel is walked linearly once (prologue, peeled last chunk, epilogue: every role ends with
vmcnt(0), so the queue is empty at the role boundaries); (2) every innermost loop that contains such loads (label .. backward
branch; forward branches of wave-uniform `if`s inside it are walked through) is replayed `ITER` times on its own, the queue carried from one iteration to the next,
which reaches the steady state the hand-derived counts are written for.  The tool also reports, per loop, the vmcnt values seen
and the deepest queue, so the expected figures (transform waves: 3 in flight / vmcnt(2); DMA waves: 16 / vmcnt(15), vmcnt(9))
can be asserted by tests/test_kernel_resources.py.

usage: python tools/asm_vm_audit.py file.s [kernel-name-substring]   -> exit status 1 on violations
"""
import re
import sys

ITER = 3


def _regs(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def _instructions(body):
    """[(kind, text)] with kind in {'label', 'ins'}; comments / directives dropped"""
    out = []
    for ln in body.split("\n"):
        s = ln.strip()
        if s.startswith(";;#ASMSTART"):
            out.append(("asm", True))
            continue
        if s.startswith(";;#ASMEND"):
            out.append(("asm", False))
            continue
        if not s or s.startswith((";", "//")):
            continue
        if re.match(r"^\.?[A-Za-z_][\w.$]*:", s):
            out.append(("label", s.split(":")[0]))
            continue
        if s.startswith("."):
            continue
        s = s.split(";")[0].strip()
        if s:
            out.append(("ins", s))
    return out


def _simulate(ins, inflight, bad, where, stats):
    in_asm = False
    for kind, s in ins:
        if kind == "asm":
            in_asm = s
            continue
        if kind != "ins":
            continue
        op = s.split()[0]
        if op.startswith("global_load_lds") or (op.startswith("buffer_load") and " lds" in s):
            used = _regs(s[len(op):])
            for rs, src in inflight:
                if rs & used:
                    bad.append((where, s, src))
            inflight.append((set(), s))
            stats["depth"] = max(stats["depth"], len(inflight))
            continue
        if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            ops = s[len(op):].split(",")
            dst, rest = _regs(ops[0]), _regs(",".join(ops[1:]))
            if not in_asm:
                dst = set()           # a load hipcc sees: it inserts its own waits; it only occupies a queue slot here
            for rs, src in inflight:
                if rs & (dst | rest):
                    bad.append((where, s, src))
            inflight.append((dst, s))
            stats["depth"] = max(stats["depth"], len(inflight))
            stats["loads"] += 1
            stats["asm_loads"] = stats.get("asm_loads", 0) + (1 if in_asm else 0)
            continue
        if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
            used = _regs(s[len(op):])
            for rs, src in inflight:
                if rs & used:
                    bad.append((where, s, src))
            inflight.append((set(), s))
            continue
        if op == "s_waitcnt":
            mm = re.search(r"vmcnt\((\d+)\)", s)
            if mm:
                n = int(mm.group(1))
                stats["waits"].append(n)
                if len(inflight) > n:
                    del inflight[:len(inflight) - n]
            continue
        if op.startswith("s_"):
            continue
        used = _regs(s[len(op):])
        for rs, src in inflight:
            if rs & used:
                bad.append((where, s, src))


def audit(path, name_filter="wino_fused_kernel"):
    text = open(path).read()
    report, bad = [], []
    for m in re.finditer(r"^(_Z\w+):[^\n]*$", text, re.M):
        name = m.group(1)
        if name_filter not in name:
            continue
        end = text.index(".Lfunc_end", m.end())
        ins = _instructions(text[m.end():end])
        # (1) linear walk of the whole kernel
        st = {"depth": 0, "loads": 0, "waits": []}
        q = []
        _simulate(ins, q, bad, name + " [linear]", st)
        report.append((name, "linear", st, len(q)))
        # (2) innermost loops with asm loads, replayed
        labels = {t: i for i, (k, t) in enumerate(ins) if k == "label"}
        for i, (k, s) in enumerate(ins):
            if k != "ins" or not s.startswith(("s_cbranch", "s_branch")):
                continue
            tgt = s.split()[-1]
            if tgt in labels and labels[tgt] < i:
                body = ins[labels[tgt]:i + 1]
                nested = False        # forward branches inside the body (a wave-uniform `if`) are walked through linearly; an inner
                for jj, (kk, tt) in enumerate(body[:-1]):       # backward branch would make this an outer loop
                    if kk == "ins" and tt.startswith(("s_cbranch", "s_branch")):
                        t2 = tt.split()[-1]
                        if t2 in labels and labels[t2] <= labels[tgt] + jj:
                            nested = True
                if nested:
                    continue
                if not any(kk == "ins" and tt.startswith("global_load_dwordx4") for kk, tt in body):
                    continue
                st = {"depth": 0, "loads": 0, "waits": []}
                q = []
                for it in range(ITER):
                    if it == ITER - 1:
                        st["waits"] = []
                    _simulate(body, q, bad, f"{name} [loop {tgt}, iteration {it}]", st)
                report.append((name, f"loop {tgt}", st, len(q)))
    return report, bad


if __name__ == "__main__":
    rep, bad = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "wino_fused_kernel")
    for name, what, st, left in rep:
        print(f"{name[:50]} {what}: {st['loads']} register loads, deepest queue {st['depth']}, vmcnt values {sorted(set(st['waits']))}, "
              f"{left} in flight at the end")
    print(f"{len(bad)} violations")
    for where, ins, src in bad[:20]:
        print(f"  {where[-40:]}: `{ins}` touches registers of in-flight `{src}`")
    sys.exit(1 if bad else 0)
