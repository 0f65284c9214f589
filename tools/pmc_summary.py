#!/usr/bin/env python3
"""Per-kernel average of the PMC counters in a rocprofv3 rocpd .db (run with --kernel-trace --pmc <COUNTER>).
Usage: python tools/pmc_summary.py <results.db> [name-substring ...]   -> JSON on stdout
Schema differences between rocprofv3 builds are handled by discovery: the pmc table is whichever
table has a 'value' column and a dispatch/event id; counter and kernel names are joined by id."""
import json, sqlite3, sys

db = sys.argv[1]
filt = sys.argv[2:]
con = sqlite3.connect(db)
cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
def cols(t): return [r[1] for r in cur.execute(f"pragma table_info('{t}')")]
out = {"_tables": {}}
# preferred: the convenience view rocprofv3 ships
view = next((t for t in tables if t.lower() in ("counters_collection", "pmc_events_view")), None)
if view is None:
    view = next((t for t in tables if "counter" in t.lower() and "value" in cols(t)), None)
out["_view"] = view
if view is None:
    out["_tables"] = {t: cols(t) for t in tables if "pmc" in t.lower() or "counter" in t.lower() or "kernel" in t.lower()}
    print(json.dumps(out, indent=1)); sys.exit(0)
c = cols(view)
kcol = next(x for x in c if x in ("kernel_name", "name", "kernel"))
ncol = next(x for x in c if x in ("counter_name", "pmc_name", "counter"))
agg = {}
for k, n, v in cur.execute(f"select {kcol}, {ncol}, value from {view}"):
    if filt and not any(f in k for f in filt):
        continue
    a = agg.setdefault((k.split("(")[0][:80], n), [0, 0.0])
    a[0] += 1; a[1] += float(v)
res = {}
for (k, n), (cnt, tot) in sorted(agg.items()):
    res.setdefault(k, {})[n] = {"dispatches": cnt, "avg": tot / cnt}
print(json.dumps(res, indent=1))
