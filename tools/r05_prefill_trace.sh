#!/bin/bash
# kernel trace of the 2048-token prefill of an 8-layer R1 rank shard (3 dense + 5 MoE): gpurun -- tools/r05_prefill_trace.sh <tag>
tag=${1:-r05_prefill}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp; rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py 8 2048 > $out/prefill.txt 2>&1
python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pp/t_results.db --last-fraction 0.3 > $out/kerneltrace_prefill_2048.txt
cat $out/prefill.txt | tail -2
head -40 $out/kerneltrace_prefill_2048.txt
