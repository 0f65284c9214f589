#!/bin/bash
# round 4, call 1: reproduce the graph replay miscompare cheaply
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call1
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python tools/graph_repro.py > $out/repro.txt 2>&1
echo "repro rc=$?" >> $out/repro.txt
tail -60 $out/repro.txt
for pred in test_gpu_xgmi test_gpu_xgmi test_gpu_w8a8 test_gpu_sampler test_gpu_llama; do
  echo "== $pred + llama reference" >> $out/pairs.txt
  timeout 400 python -m pytest tests/$pred.py tests/test_llama_reference.py -m gpu -q --timeout 300 2>&1 | tail -15 >> $out/pairs.txt
done
grep -n "==\|passed\|failed\|graph rows" $out/pairs.txt | cut -c1-400
