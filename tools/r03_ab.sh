#!/bin/bash
# Same-box A/B of the in-tree library against build_probe/lib_base.so: kernel-time averages inside a graph-replayed bench step
# (rocprofv3 kernel trace) at the batch sizes given, then ms/step of both, twice.   tools/r03_ab.sh <tag> <pattern> [bs...]
tag=$1; pat=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for bs in "$@"; do
  for lib in build_probe/lib_base.so ""; do
    L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
    rm -rf /tmp/pa
    CHITU_HIP_LIB=$L rocprofv3 --kernel-trace --stats -d /tmp/pa -o t -- python $GRAFT_REPO_ROOT/bench.py --bs $bs --steps 8 --warmup 2 --no-bs1 --no-llama --no-cpu-baseline --no-roofline > /tmp/pa.log 2>&1
    echo "== bs $bs ${lib:-in-tree}" >> $out/kernel_time.txt
    python $GRAFT_REPO_ROOT/tools/step_breakdown.py /tmp/pa/t_results.db 8 | head -3 >> $out/kernel_time.txt
    python $GRAFT_REPO_ROOT/tools/step_breakdown.py /tmp/pa/t_results.db 8 | grep -E "$pat" | cut -c1-120 >> $out/kernel_time.txt
  done
done
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in build_probe/lib_base.so ""; do
    L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
    CHITU_HIP_LIB=$L python bench.py --no-cpu-baseline --no-roofline --no-llama 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${lib:-in-tree}', 'bs16', d['ms_per_step'], 'bs1', d.get('bs1',{}).get('ms_per_step'), 'bs32', d.get('bs32',{}).get('ms_per_step'))" >> $out/ms_per_step.txt
  done
done
cat $out/kernel_time.txt $out/ms_per_step.txt
