#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_mixtral; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_mixtral.py tests/test_gpu_moe_int8.py tests/test_gpu_w8a8.py tests/test_gpu_llama.py -x -q > $out/tests4.txt 2>&1; echo "tests rc=$?"; tail -4 $out/tests4.txt
for i in 1 2; do
  echo "== + top-2 sum inside the next residual add" | tee -a $out/line4.txt
  timeout 300 python tools/run_extra.py mixtral 16 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: (v['ms_per_step'], v['roofline_frac']) for k, v in d.items() if k.startswith('bs')})" | tee -a $out/line4.txt
done
timeout 200 python tools/run_extra.py llama 16 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('llama', {k: (v['ms_per_step'], v['roofline_frac']) for k, v in d.items() if k.startswith('bs')})" | tee -a $out/line4.txt
