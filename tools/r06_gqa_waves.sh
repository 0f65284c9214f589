#!/bin/bash
# Llama-3-8B decode step against the GQA decode launch's waves-per-CU target (splits = target * CUs / (batch * kv_heads)):  gpurun -- bash tools/r06_gqa_waves.sh
cd $GRAFT_REPO_ROOT
for w in 4 8 16 4 8 16 2; do
  echo -n "waves_per_cu=$w "
  CHITU_GQA_WAVES_PER_CU=$w CHITU_BENCH_EXTRA_BATCHES=1,4,16 timeout 300 python tools/run_extra.py llama 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(v['ms_per_step'],v['roofline_frac']) for k,v in d.items() if k.startswith('bs')})"
done
