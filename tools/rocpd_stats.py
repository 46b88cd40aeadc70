#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (kernel-trace) as a per-kernel table: calls, total/avg/min/max us, % of GPU time.
Usage: python tools/rocpd_stats.py gpurun_out/prof/xxx_results.db [out.md]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for n, s, e in rows:
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\((?!.*<).*$", "", n)          # drop the trailing argument list
    a = agg.setdefault(n, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append(f"| {n[:90]} | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.1f} | {a[2] / 1e3:.1f} | {a[3] / 1e3:.1f} | {100 * a[1] / tot:.1f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
