#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_dbg
mkdir -p $out
cd $GRAFT_REPO_ROOT
for i in 1 2; do
timeout 900 python -m pytest tests/test_gpu_xgmi.py tests/test_llama_reference.py -q 2>&1 | grep -E "^E  |passed|failed|FAILED|skipped" | cut -c1-600 | head -12 >> $out/dbg6.txt
done
cat $out/dbg6.txt
