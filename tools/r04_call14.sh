#!/bin/bash
# round 4, call 14: the whole GPU suite once more on the final tree, then the round sweep
head=$1
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call14
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1300 python -m pytest tests -m gpu -q --timeout 600 > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
grep -n "GRAPH CAPTURE\|graph mismatch probe\|hipGraph captures\|passed\|failed\|rc=" $out/tests.txt | cut -c1-2000
timeout 600 bash tools/round_sweep.sh r04_final2 $head > $out/sweep.log 2>&1
tail -2 $out/sweep.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_final2/bench.json"))
print({k:d.get(k) for k in ("value","ms_per_step","step_roofline_frac","invalid")})
print("bs1",d.get("bs1"),"bs32",d.get("bs32"))
print("graph_verified",d.get("graph_verified"))
print("roofline_kernels",[ (k["kernel"][:24],k["frac"],k["avg_launch_us"]) for k in d.get("roofline_kernels") or []])
for k in ("llama3_8b","v2_lite","mixtral_8x7b_int8","ep8_rank"): print(k, {b:v for b,v in d.get(k,{}).items() if b.startswith("bs")})
print(d["cpu_baseline"].get("reference_fields"))
PY
