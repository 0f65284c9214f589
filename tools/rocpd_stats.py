#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel calls / total / avg / min / max (us).
Usage: python tools/rocpd_stats.py <results.db> [--last-fraction F] > profiles/<name>.txt"""
import sqlite3
import sys

db = sys.argv[1]
frac = float(sys.argv[sys.argv.index("--last-fraction") + 1]) if "--last-fraction" in sys.argv else 1.0
con = sqlite3.connect(db)
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name_col}, start, end from kernels order by start").fetchall()
if frac < 1.0:
    rows = rows[int(len(rows) * (1 - frac)):]
agg = {}
for n, s, e in rows:
    d = (e - s) / 1e3
    a = agg.setdefault(n, [0, 0.0, 1e30, 0.0])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values())
span = (rows[-1][2] - rows[0][1]) / 1e3 if rows else 0
print(f"# {db}: {len(rows)} dispatches, sum of kernel time {tot/1e3:.3f} ms, wall span {span/1e3:.3f} ms, "
      f"busy {100*tot/max(span,1e-9):.1f}%")
print(f"{'calls':>8} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    short = n if len(n) < 110 else n[:107] + "..."
    print(f"{a[0]:8d} {a[1]:12.1f} {a[1]/a[0]:10.2f} {a[2]:10.2f} {a[3]:10.2f} {100*a[1]/tot:6.2f}  {short}")
