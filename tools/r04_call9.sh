#!/bin/bash
# round 4, call 9: the whole GPU suite on the final tree, then the round sweep (bench line, step breakdowns, PMC passes)
head=$1
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call9
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1300 python -m pytest tests -m gpu -q --timeout 600 > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
grep -n "GRAPH CAPTURE\|graph mismatch probe\|hipGraph captures\|passed\|failed\|rc=" $out/tests.txt | cut -c1-2000
timeout 600 bash tools/round_sweep.sh r04_final $head > $out/sweep.log 2>&1
tail -2 $out/sweep.log
head -c 600 gpurun_out/r04_final/bench.json; echo
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_final/bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step","step_roofline_frac","invalid")})
print("bs1",d.get("bs1"),"bs32",d.get("bs32"))
print("graph_verified",d.get("graph_verified"))
print("roofline_kernels",d.get("roofline_kernels"))
for k in ("llama3_8b","v2_lite","mixtral_8x7b_int8","ep8_rank"): print(k, d.get(k))
PY
head -16 gpurun_out/r04_final/step_breakdown_bs1.txt | cut -c1-150
cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo -n "v2lite CHITU_FUSE_ATTN_NORM_MAX_BS=$v: " >> $out/v2lite_ab.txt
  CHITU_FUSE_ATTN_NORM_MAX_BS=$v timeout 200 python tools/run_extra.py v2lite 32 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['bs1']['ms_per_step'], d['bs16']['ms_per_step'])" >> $out/v2lite_ab.txt
done
cat $out/v2lite_ab.txt
