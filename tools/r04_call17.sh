#!/bin/bash
# round 4, call 17: kernel traces of the two extras whose roofline fractions are lowest (V2-Lite, the EP rank), for the next round
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call17
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pv; timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/pv -o t -- python $GRAFT_REPO_ROOT/tools/run_extra.py v2lite 16 > $out/v2lite.log 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pv/t_results.db --last-fraction 0.3 > $out/kerneltrace_v2lite.txt
head -22 $out/kerneltrace_v2lite.txt | cut -c1-170
