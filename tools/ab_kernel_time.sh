#!/bin/bash
# Kernel-time A/B on ONE box: tools/ab_kernel_time.sh <kernel-name-substring> <bs> <lib|""> [<lib|""> ...]
# For every library ("" = the in-tree build) a rocprofv3 kernel trace of `bench.py --bs B --steps 8` is taken and the
# matching kernels' calls / total / avg / min / max us are printed (last 70 % of the dispatches: the graph replays).
# Far more sensitive than ms/step for one kernel (step time on a box moves by +-1 %, a kernel's average by +-0.1 us).
pat=$1; bs=$2; shift 2
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for lib in "$@"; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  rm -rf /tmp/pa
  CHITU_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pa -o t -- python $GRAFT_REPO_ROOT/bench.py --bs $bs --steps 8 --warmup 2 --no-bs1 --no-llama --no-cpu-baseline --no-roofline --no-calibration > /tmp/pa.log 2>&1
  echo "== ${lib:-in-tree}"
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pa/t_results.db --last-fraction 0.7 | grep -E "$pat" | cut -c1-100
done; done
