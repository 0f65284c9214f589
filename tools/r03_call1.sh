#!/bin/bash
# round-3 GPU call 1: hand-off / barrier probes, the tests touched by the advisor fixes, a reference bench line
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call1
mkdir -p $out
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_gridbar.hip -o /tmp/probe_gridbar.bin 2>$out/build_err.txt
timeout 120 /tmp/probe_gridbar.bin > $out/probe_gridbar.txt 2>&1
timeout 180 tools/probe_handoff.bin > $out/probe_handoff.txt 2>&1
echo "probe rc=$?" >> $out/probe_handoff.txt
timeout 600 python -m pytest tests/test_gpu_gqa.py tests/test_gpu_xgmi.py tests/test_gpu_mla.py -x -q > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
timeout 300 python bench.py --no-cpu-baseline --no-llama > $out/bench.json 2> $out/bench_err.txt
tail -5 $out/probe_handoff.txt; tail -3 $out/tests.txt
