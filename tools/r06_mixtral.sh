#!/bin/bash
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_mixtral; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_mixtral.py tests/test_gpu_moe_int8.py -x -q > $out/tests2.txt 2>&1; echo "tests rc=$?"; tail -3 $out/tests2.txt
for opt in "" "moe_i8_wk=8" "moe_i8_wk=1" ""; do
  echo "== new heuristic; CHITU_DEBUG_OPTIONS=$opt" | tee -a $out/line2.txt
  CHITU_DEBUG_OPTIONS=$opt timeout 300 python tools/run_extra.py mixtral 16 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: (v['ms_per_step'], v['roofline_frac']) for k, v in d.items() if k.startswith('bs')})" | tee -a $out/line2.txt
done
