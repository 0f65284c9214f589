#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call6
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mla.py tests/test_gpu_deepseek.py -x -q > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
tail -5 $out/tests.txt
tools/r03_ab.sh r03_call6 "mla_|absorb" 16 1
