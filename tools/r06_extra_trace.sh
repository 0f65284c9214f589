#!/bin/bash
# Kernel trace of one extra workload's decode step at one batch size:  gpurun -- bash tools/r06_extra_trace.sh <llama|v2lite|mixtral> <bs> [more "<model> <bs>" pairs]
# -> gpurun_out/r06_extra/kerneltrace_<model>_bs<bs>.txt (the last 40 % of the dispatches of a 24-step run = graph replays of the timed steps)
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_extra; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
while [ $# -ge 2 ]; do
  m=$1; bs=$2; shift 2
  rm -rf /tmp/pe_$m$bs
  CHITU_BENCH_EXTRA_BATCHES=$bs timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pe_$m$bs -o t -- python $GRAFT_REPO_ROOT/tools/run_extra.py $m 24 > $out/$m$bs.log 2>&1
  tail -1 $out/$m$bs.log | cut -c1-300
  timeout 60 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pe_$m$bs/t_results.db --last-fraction 0.4 > $out/kerneltrace_${m}_bs$bs.txt
  head -24 $out/kerneltrace_${m}_bs$bs.txt | cut -c1-190
done
