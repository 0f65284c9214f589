#!/bin/bash
# FETCH_SIZE of the prefill expert GEMMs (5-layer R1 rank shard, 2048 tokens) under 64- and 128-slot tiles: one --pmc pass each.
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_moe128; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for bm in 64 128; do
  rm -rf /tmp/pf_$bm
  CHITU_MOE_TILED_BLOCK_M=$bm timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf_$bm -o pmc -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py 5 2048 > $out/pmc_fetch_$bm.log 2>&1
  db=$(ls /tmp/pf_$bm/*.db /tmp/pf_$bm/*/*.db 2>/dev/null | head -1)
  echo "== block_m $bm: FETCH_SIZE avg per launch (KB as reported; x2 on gfx950 for wide streaming reads)" | tee -a $out/pmc_fetch.txt
  [ -n "$db" ] && timeout 60 python $GRAFT_REPO_ROOT/tools/pmc_summary.py $db moe_gemm_tiled | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items():
    if k.startswith('_'): continue
    f=v.get('FETCH_SIZE',{})
    print(k[:70], 'dispatches', f.get('dispatches'), 'FETCH_SIZE avg KB', round(f.get('avg',0),1), '-> x2 =', round(f.get('avg',0)*2/1e6,3), 'GB')" | tee -a $out/pmc_fetch.txt
done
