#!/bin/bash
# N = 2 functional run (exercises enable_xgmi's two-shot self-test) + MLA split-count sweep on the step
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call13
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --gpus 2 --layers 12 --steps 16 --warmup 4 --no-llama --no-cpu-baseline > $out/bench_n2.json 2> $out/bench_n2_err.txt
echo "n2 rc=$?"; grep -c xgmi $out/bench_n2.json; grep "chitu_amd\]" $out/bench_n2_err.txt $out/bench_n2.json | head
timeout 300 python -m pytest tests/test_gpu_xgmi.py -x -q -k "two_processes or process" > $out/xgmi_tests.txt 2>&1; tail -3 $out/xgmi_tests.txt
for bs in 16 1 32; do
  for s in default 4 8 11 16 22; do
    if [ $s = default ]; then unset CHITU_MLA_SPLITS; else export CHITU_MLA_SPLITS=$s; fi
    r=$(timeout 300 python bench.py --bs $bs --steps 32 --warmup 4 --no-bs1 --no-llama --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "bs $bs splits $s: $r ms/step" | tee -a $out/mla_splits.txt
  done
done
