#!/bin/bash
# round 4, call 5: uncached-recycling repro without graphs; the two norm-prologue fusions; comm parking
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call5
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 python tools/uc_alias_repro.py > $out/uc_alias.txt 2>&1; echo "rc=$?" >> $out/uc_alias.txt
grep -v amdgpu.ids $out/uc_alias.txt | tail -20
timeout 900 python -m pytest tests/test_gpu_deepseek.py tests/test_gpu_graphs.py tests/test_gpu_llama.py -m gpu -q --timeout 300 -x > $out/tests_a.txt 2>&1; echo "rc=$?" >> $out/tests_a.txt
tail -25 $out/tests_a.txt | cut -c1-400
timeout 600 python -m pytest tests/test_gpu_xgmi.py -m gpu -q --timeout 300 -k "two_ranks or all_gather_calls or missing_peer or unsupported or tp4" > $out/tests_b.txt 2>&1; echo "rc=$?" >> $out/tests_b.txt
tail -8 $out/tests_b.txt | cut -c1-400
