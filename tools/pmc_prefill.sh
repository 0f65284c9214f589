#!/bin/bash
# PMC counter passes over the 2048-token prefill of a 5-layer R1 rank shard (3 dense + 2 MoE), one rocprofv3 run per counter
# group (--kernel-trace --pmc only):  gpurun -- tools/pmc_prefill.sh <tag>   -> gpurun_out/<tag>/pmc_*.json + summary.txt
tag=${1:-pmc_prefill}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  t=$(echo "$grp" | tr ' ' '+' | cut -c1-40)
  rm -rf "/tmp/pmc_$t"
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "/tmp/pmc_$t" -o pmc -- python "$GRAFT_REPO_ROOT/tools/prefill_bench.py" 5 2048 > "$out/pmc_$t.log" 2>&1
  db=$(ls /tmp/pmc_$t/*.db /tmp/pmc_$t/*/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then
    python "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$db" chitu:: > "$out/pmc_$t.json" 2>> "$out/pmc_$t.log"
    python "$GRAFT_REPO_ROOT/tools/rocpd_stats.py" "$db" > "$out/pmc_$t.durations.txt" 2>/dev/null
  else
    echo "no db for $grp" >> "$out/pmc_$t.log"
  fi
done
python - "$out" <<'PY' | tee "$out/summary.txt"
import glob, json, os, sys
d = sys.argv[1]
tabs = {}
for f in sorted(glob.glob(os.path.join(d, "pmc_*.json"))):
    try:
        tabs[os.path.basename(f)] = json.load(open(f))
    except Exception as e:
        print("bad", f, e)
kernels = set()
for t in tabs.values():
    kernels |= {k for k in t if not k.startswith("_")}
for k in sorted(kernels):
    if not any(s in k for s in ("mla_prefill", "moe_gemm_tiled", "fp8_gemm_tiled", "bf16_gemm_tiled", "absorb", "moe_align")):
        continue
    row = {}
    for t in tabs.values():
        for c, v in t.get(k, {}).items():
            row[c] = v.get("avg")
    print(k[:70], json.dumps({c: (round(v, 1) if isinstance(v, float) else v) for c, v in row.items()}))
PY
