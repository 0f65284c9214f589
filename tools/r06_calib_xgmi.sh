#!/bin/bash
# round 6 items 8 + 9: the default bench line (tail_launch_us / static facts in box_calibration), the xGMI process test with the
# pre-flight report, and the N = 2 functional line on one shared GPU (both ranks on cuda:0, 12 layers) with `xgmi_enable` in it.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_calib; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_xgmi.py -x -q -k "ranks_as_processes or two_ranks_one_process" > $out/xgmi_tests.txt 2>&1; echo "rc=$?" >> $out/xgmi_tests.txt; tail -4 $out/xgmi_tests.txt
timeout 600 python bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r06_calib/bench_default.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d.get("bs1", {}).get("ms_per_step"), d.get("bs32", {}).get("ms_per_step"))
c = d["box_calibration"]
print(json.dumps({k: c[k] for k in ("tail_launch_us", "tail_launch_vs_reference_box", "static_device_facts", "step_time_factor") if k in c}, indent=1)[:3000])
PY
timeout 900 python bench.py --gpus 2 --layers 12 --steps 16 --warmup 4 --no-llama --no-cpu-baseline > $out/bench_n2_shared_gpu_12layers.json 2> $out/bench_n2.err; echo "n2 rc=$?"
python - <<'PY'
import json
lines = [l for l in open("gpurun_out/r06_calib/bench_n2_shared_gpu_12layers.json").read().splitlines() if l.startswith("{")]
d = json.loads(lines[-1])
print(d["ms_per_step"], json.dumps(d["collectives"].get("xgmi_enable"))[:1500], d["collectives"].get("graph_form"))
PY
