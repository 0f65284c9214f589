#!/usr/bin/env python3
"""Per-decode-step kernel breakdown from a rocprofv3 rocpd .db of `bench.py --steps K`.
Usage: python tools/step_breakdown.py <db> <timed_steps> [moe_layers_per_step=58] [tail_markers=2*moe_layers]
The step boundaries are found from the routing launches (one `gate_route*` kernel per MoE layer per step);
tail_markers = routing launches bench.py issues AFTER the timed steps (roofline leg: two eager
routing-capture steps); run bench.py with --no-bs1 --no-llama."""
import sqlite3, sys
db, steps = sys.argv[1], int(sys.argv[2])
nmoe = int(sys.argv[3]) if len(sys.argv) > 3 else 58
con = sqlite3.connect(db)
rows = con.execute("select name, start, end from kernels order by start").fetchall()
al = [i for i, r in enumerate(rows) if "gate_route" in r[0]]
tail = int(sys.argv[4]) if len(sys.argv) > 4 else 2 * nmoe
first = al[-tail - nmoe * steps]
t0 = rows[first][1]
t1 = rows[al[-tail]][1] if tail > 0 else rows[-1][2] + 1  # tail 0 (bench.py --no-roofline): up to the last dispatch
sel = [r for r in rows if t0 <= r[1] < t1]
agg = {}
for n, s, e in sel:
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += (e - s) / 1e3
tot = sum(a[1] for a in agg.values())
print(f"# {db}: {steps} timed steps; wall {(t1-t0)/1e6/steps:.3f} ms/step (profiled); kernel-time sum {tot/1e3/steps:.3f} ms/step; "
      f"{len(sel)/steps:.0f} dispatches/step")
print(f"{'calls/step':>10} {'us/step':>10} {'avg_us':>8} {'pct':>6}  kernel")
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{a[0]/steps:10.1f} {a[1]/steps:10.1f} {a[1]/a[0]:8.2f} {100*a[1]/tot:6.2f}  {n[:120]}")
