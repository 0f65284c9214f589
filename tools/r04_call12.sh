#!/bin/bash
# round 4, call 12: KV split count of the GQA decode launch at bs 1 / 16 (Llama-3-8B), one box
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call12
mkdir -p $out
cd $GRAFT_REPO_ROOT
for s in 64 32 16 8; do
  echo "== CHITU_GQA_MAX_SPLITS=$s" >> $out/splits.txt
  CHITU_GQA_MAX_SPLITS=$s timeout 200 python tools/llama_ab.py --bs 1,16 --reps 1 --steps 40 2>/dev/null | grep ms_per_step >> $out/splits.txt
done
cat $out/splits.txt
