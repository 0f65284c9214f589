#!/bin/bash
# PMC counter passes over the decode step's kernels (north_star: "rocprof HBM GB/s and MFMA utilisation").
#   tools/pmc_passes.sh <out_dir> [bench.py args...]
# One rocprofv3 run per counter group (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, PMC slots),
# --kernel-trace --pmc only (no sys/hip/memory tracing next to counters).  The step runs EAGERLY (--no-graph) on a
# few layers so every dispatch is attributed to its kernel; tools/pmc_report.py turns the .db files into
# profiles/r02_pmc_step.json.
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--layers 8 --steps 2 --warmup 1 --no-graph --no-bs1 --no-llama --no-cpu-baseline --no-roofline --no-calibration $*"
rocprofv3 -L > "$out/counters_available.txt" 2>&1
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
           "SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU"; do
  tag=$(echo "$grp" | tr ' ' '+' | cut -c1-40)
  rm -rf "/tmp/pmc_$tag"
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "/tmp/pmc_$tag" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" $args > "$out/pmc_$tag.log" 2>&1
  db=$(ls /tmp/pmc_$tag/*.db /tmp/pmc_$tag/*/*.db 2>/dev/null | head -1)
  if [ -n "$db" ]; then
    python "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$db" chitu:: > "$out/pmc_$tag.json" 2>> "$out/pmc_$tag.log"
    python "$GRAFT_REPO_ROOT/tools/rocpd_stats.py" "$db" > "$out/pmc_$tag.durations.txt" 2>/dev/null
  else
    echo "no db for $grp" >> "$out/pmc_$tag.log"
  fi
done
ls -la "$out"
