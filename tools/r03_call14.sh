#!/bin/bash
# XCD-blocked tile order of the tiled prefill GEMM vs the row-major order (build_probe/lib_tiled_rowmajor.so), one box
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call14
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_fp8.py -x -q -k "tiled" > $out/tests.txt 2>&1; tail -2 $out/tests.txt
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for lib in "" build_probe/lib_tiled_rowmajor.so; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  rm -rf /tmp/pp
  CHITU_HIP_LIB=$L rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py 8 > /tmp/pp.log 2>&1
  echo "== ${lib:-in-tree (XCD-blocked)}" | tee -a $out/ab.txt
  grep prompt_tokens /tmp/pp.log | tee -a $out/ab.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pp/t_results.db --last-fraction 0.4 | grep -E "fp8_gemm_tiled|moe_gemm_tiled|bf16_gemm_tiled" | cut -c1-110 | tee -a $out/ab.txt
done; done
