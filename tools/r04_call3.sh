#!/bin/bash
# round 4, call 3: new graph tests alone, then the whole GPU suite (durations listed)
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call3
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_graphs.py tests/test_llama_reference.py -m gpu -q --timeout 300 > $out/new_tests.txt 2>&1
echo "rc=$?" >> $out/new_tests.txt
tail -30 $out/new_tests.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=30 > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
grep -n "GRAPH CAPTURE\|graph mismatch probe\|hipGraph captures\|passed\|failed\|rc=" $out/tests.txt | cut -c1-3000
grep -A40 "slowest" $out/tests.txt | head -45
