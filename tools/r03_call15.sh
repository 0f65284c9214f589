#!/bin/bash
# prefill expert GEMM2: four 128-row tiles per workgroup through one pipeline (in-tree) vs a workgroup per tile
# (build_probe/lib_moe_tiled_nrep1.so), one box
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call15
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_moe.py -x -q -k "tiled" > $out/tests.txt 2>&1; tail -2 $out/tests.txt
cd /tmp && export TMPDIR=/tmp
for lib in "" build_probe/lib_moe_tiled_nrep1.so ""; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  rm -rf /tmp/pp
  CHITU_HIP_LIB=$L rocprofv3 --kernel-trace --stats -d /tmp/pp -o t -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py 8 > /tmp/pp.log 2>&1
  echo "== ${lib:-in-tree (4 tiles per workgroup)}" | tee -a $out/ab.txt
  grep prompt_tokens /tmp/pp.log | tee -a $out/ab.txt
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pp/t_results.db --last-fraction 0.4 | grep -E "moe_gemm_tiled" | cut -c1-110 | tee -a $out/ab.txt
done
