#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call11
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for mode in before now; do
  if [ $mode = before ]; then export CHITU_MOE_FUSE_SILU=0; else unset CHITU_MOE_FUSE_SILU; fi
  rm -rf /tmp/prof_$mode
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o t -- python $GRAFT_REPO_ROOT/tools/run_extra.py v2lite 8 > $out/run_$mode.txt 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/prof_$mode/t_results.db --last-fraction 0.3 > $out/v2lite_${mode}_kerneltrace.txt
done
