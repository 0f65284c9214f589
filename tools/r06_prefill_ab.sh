#!/bin/bash
# Round 6 item 4: MX-scaled K=128 fp8 MFMA in the tiled GEMMs -- parity tests on the MX build, then the bench's prefill section
# (layer ms, per-shape GEMM TFLOP/s) under both builds on the same box, twice.  build_probe/lib_fp8_nomx.so = -DCHITU_FP8_MX=0.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_mx; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_fp8.py tests/test_gpu_moe.py tests/test_gpu_production_shapes.py -x -q > $out/tests.txt 2>&1; echo "rc=$?" >> $out/tests.txt; tail -4 $out/tests.txt
timeout 600 python -m pytest tests/test_gpu_deepseek.py -x -q -k "prefill" > $out/tests_prefill.txt 2>&1; echo "rc=$?" >> $out/tests_prefill.txt; tail -3 $out/tests_prefill.txt
for rep in 1 2; do for lib in "" build_probe/lib_fp8_nomx.so; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  echo "== ${lib:-in-tree (MX)}" >> $out/ab.txt
  CHITU_HIP_LIB=$L timeout 300 python -c "
import json, bench
print(json.dumps(bench.prefill_extra(2048)))" 2>/dev/null | tail -1 >> $out/ab.txt
  CHITU_HIP_LIB=$L timeout 300 python tools/prefill_bench.py 8 2048 8192 2>/dev/null | grep prompt_tokens >> $out/ab.txt
done; done
cat $out/ab.txt
