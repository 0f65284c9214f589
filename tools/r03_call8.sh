#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call8
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_mla.py tests/test_gpu_deepseek.py -x -q -k "tile_major or merge_uv_quant_tile" > $out/tests.txt 2>&1
echo "tests rc=$?" >> $out/tests.txt
tail -5 $out/tests.txt
cd /tmp && export TMPDIR=/tmp
for bs in 16 32; do
  for tm in 0 1; do
    rm -rf /tmp/pa
    CHITU_TILE_MAJOR=$tm rocprofv3 --kernel-trace --stats -d /tmp/pa -o t -- python $GRAFT_REPO_ROOT/bench.py --bs $bs --steps 8 --warmup 2 --no-bs1 --no-llama --no-cpu-baseline --no-roofline > /tmp/pa.log 2>&1
    echo "== bs $bs CHITU_TILE_MAJOR=$tm" >> $out/kernel_time.txt
    python $GRAFT_REPO_ROOT/tools/step_breakdown.py /tmp/pa/t_results.db 8 | head -1 >> $out/kernel_time.txt
    python $GRAFT_REPO_ROOT/tools/step_breakdown.py /tmp/pa/t_results.db 8 | grep -E "fp8_gemm|rmsnorm|merge_uv" | cut -c1-120 >> $out/kernel_time.txt
  done
done
cat $out/kernel_time.txt
