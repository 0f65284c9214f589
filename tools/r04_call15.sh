#!/bin/bash
# round 4, call 15: the round sweep on the final tree (call 14 landed on a box whose GPU faulted at the first kernel of
# every process -- "Memory access fault by GPU node-2", pytest and bench alike; a fresh box ran the same tree clean)
head=$1
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call15
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 420 bash tools/round_sweep.sh r04_final2 $head > $out/sweep.log 2>&1
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
