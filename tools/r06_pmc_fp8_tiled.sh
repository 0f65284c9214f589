#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_pmc_fp8; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo "$grp" | tr ' ' '+' | cut -c1-30)
  rm -rf /tmp/pg_$t
  timeout 120 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pg_$t -o pmc -- python $GRAFT_REPO_ROOT/tools/fp8_tiled_only.py > $out/$t.log 2>&1
  db=$(ls /tmp/pg_$t/*.db /tmp/pg_$t/*/*.db 2>/dev/null | head -1)
  [ -n "$db" ] && timeout 60 python - "$db" <<'PY' | tee -a $out/summary.txt
import sqlite3, sys, json
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
view = next((t for t in tables if t.lower() in ("counters_collection", "pmc_events_view")), None)
cols = [r[1] for r in cur.execute(f"pragma table_info('{view}')")]
kcol = next(x for x in cols if x in ("kernel_name", "name", "kernel")); ncol = next(x for x in cols if x in ("counter_name", "pmc_name", "counter"))
gcol = next((x for x in cols if x in ("grid_size", "grid_size_x", "grid_x")), None)
agg = {}
q = f"select {kcol}, {ncol}, value" + (f", {gcol}" if gcol else "") + f" from {view}"
for row in cur.execute(q):
    k, n, v = row[0], row[1], row[2]
    if "fp8_gemm_tiled" not in k: continue
    key = (row[3] if gcol else 0, n)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += float(v)
for (g, n), (c, t) in sorted(agg.items()):
    print(f"grid {g} {n} dispatches {c} avg {t/c:.1f}")
PY
done
