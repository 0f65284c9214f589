#!/bin/bash
# round 4, call 6: same-box A/B of the two norm-prologue fusions (bs 1, bs 2), the round sweep, then the whole GPU suite
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call6
mkdir -p $out
cd $GRAFT_REPO_ROOT
B="python bench.py --no-bs1 --no-llama --no-cpu-baseline --no-roofline --steps 64 --warmup 8"
for rep in 1 2; do
  for bs in 1 2; do
    for cfg in "0 0" "1 0" "0 2" "1 2"; do
      set -- $cfg
      echo -n "bs=$bs attn_norm_fuse=$1 router_norm_fuse=$2 : " >> $out/ab.txt
      CHITU_FUSE_ATTN_NORM_MAX_BS=$1 CHITU_FUSE_ROUTER_NORM_MAX_BS=$2 $B --bs $bs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['graph_verified']['all_equal_eager'])" >> $out/ab.txt
    done
  done
done
cat $out/ab.txt
bash tools/round_sweep.sh r04_sweep1 $1 > $out/sweep.log 2>&1
tail -3 $out/sweep.log
head -c 1500 gpurun_out/r04_sweep1/bench.json; echo
head -20 gpurun_out/r04_sweep1/step_breakdown_bs1.txt | cut -c1-150
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
grep -n "GRAPH CAPTURE\|graph mismatch probe\|hipGraph captures\|passed\|failed\|rc=" $out/tests.txt | cut -c1-3000
