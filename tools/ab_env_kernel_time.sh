#!/bin/bash
# Kernel-time A/B of two ENVIRONMENT settings on ONE box (same library):
#   tools/ab_env_kernel_time.sh <kernel-name-regex> <bs> <"VAR=value"|""> [<"VAR=value"|""> ...]
# Like tools/ab_kernel_time.sh, for switches that live above the C-ABI (e.g. CHITU_Q_PROJ_FUSED=0).
pat=$1; bs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for kv in "$@"; do
  rm -rf /tmp/pa
  env $kv timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa -o t -- python $GRAFT_REPO_ROOT/bench.py --bs $bs --steps 8 --warmup 2 --no-bs1 --no-llama --no-cpu-baseline --no-roofline --no-calibration > /tmp/pa.log 2>&1
  echo "== ${kv:-default}  $(grep -o '"ms_per_step": [0-9.]*' /tmp/pa.log | head -1)"
  python $GRAFT_REPO_ROOT/tools/step_breakdown.py /tmp/pa/t_results.db 8 58 0 | grep -E "kernel-time sum|$pat" | cut -c1-110
done; done
