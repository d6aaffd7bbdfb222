"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite .db or *_kernel_trace.csv) into a small
per-kernel table (calls, total ms, avg/min/max us, share) for profiles/.
usage: python tools/rocprof_summary.py <results.db | kernel_trace.csv> <out.md> [title]"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    con = sqlite3.connect(path)
    return con.execute("select name, start, end from kernels").fetchall()


def rows_from_csv(path):
    out = []
    with open(path) as f:
        for r in csv.DictReader(f):
            out.append((r.get("Kernel_Name") or r.get("Name"), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return out


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else src
    rows = rows_from_db(src) if src.endswith(".db") else rows_from_csv(src)
    agg = defaultdict(list)
    for name, s, e in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name) if not name.startswith("void conv_igemm") and "<" not in name else re.sub(r"\((?!.*<).*$", "", name)
        agg[name[:110]].append((e - s) / 1e3)
    tot = sum(sum(v) for v in agg.values())
    with open(dst, "w") as f:
        f.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats` ; total kernel time {tot / 1e3:.2f} ms over {len(rows)} dispatches\n\n")
        f.write("| kernel | calls | total ms | share | avg us | min us | max us |\n|---|---:|---:|---:|---:|---:|---:|\n")
        for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            f.write(f"| `{name}` | {len(v)} | {sum(v) / 1e3:.3f} | {100 * sum(v) / tot:.1f}% | {sum(v) / len(v):.1f} | {min(v):.1f} | {max(v):.1f} |\n")
    print("wrote", dst)


if __name__ == "__main__":
    main()
