#!/bin/bash
# Kernel traces of the bs-1 decode steps of DeepSeek-V2-Lite (config 3) and Mixtral-8x7B int8 (config 4): the per-launch table behind the
# floor argument of docs/design/10_small_model_floors.md.  The last 40 % of the dispatches of a 24-step run = graph replays of the timed steps.
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_small; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for m in v2lite mixtral; do
  rm -rf /tmp/ps_$m
  CHITU_BENCH_EXTRA_BATCHES=1 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ps_$m -o t -- python $GRAFT_REPO_ROOT/tools/run_extra.py $m 24 > $out/$m.log 2>&1
  tail -1 $out/$m.log | cut -c1-400
  timeout 60 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/ps_$m/t_results.db --last-fraction 0.4 > $out/kerneltrace_${m}_bs1.txt
  head -30 $out/kerneltrace_${m}_bs1.txt | cut -c1-170
done
