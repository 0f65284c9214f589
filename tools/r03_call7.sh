#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call7
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_moe_align.py tests/test_gpu_mla.py tests/test_gpu_deepseek.py -x -q > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
tail -5 $out/tests.txt
cd /tmp && export TMPDIR=/tmp
for bs in 16 1; do
  for opt in "gate_small_sort=0" ""; do
    O=""; [ -n "$opt" ] && O="--opt $opt"
    rm -rf /tmp/pa
    rocprofv3 --kernel-trace --stats -d /tmp/pa -o t -- python $GRAFT_REPO_ROOT/bench.py --bs $bs --steps 8 --warmup 2 --no-bs1 --no-llama --no-cpu-baseline --no-roofline $O > /tmp/pa.log 2>&1
    echo "== bs $bs ${opt:-default}" >> $out/kernel_time.txt
    python $GRAFT_REPO_ROOT/tools/step_breakdown.py /tmp/pa/t_results.db 8 | head -1 >> $out/kernel_time.txt
    python $GRAFT_REPO_ROOT/tools/step_breakdown.py /tmp/pa/t_results.db 8 | grep -E "gate_route|mla_" | cut -c1-120 >> $out/kernel_time.txt
  done
done
cat $out/kernel_time.txt
