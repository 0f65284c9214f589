#!/bin/bash
# the prefill expert GEMMs with an L2 warm-up of the weight stream WARM steps ahead (in-tree: 3; build_probe/lib_mwarm{0,2,5}.so =
# tools/build_variant.sh mwarm<N> moe_tiled.hip -DCHITU_MOE_TILED_WARM=<N>; 0 = off), same box:
#   gpurun -- bash tools/r06_moe_warm_ab.sh    -> prefill_bench lines (8 layers of the R1 rank shard) + the expert GEMMs' kernel time at 2048 tokens
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_moe_warm; mkdir -p $out; rm -f $out/ab.txt
for rep in 1 2; do for lib in build_probe/lib_mwarm0.so build_probe/lib_mwarm2.so "" build_probe/lib_mwarm5.so; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  echo "== ${lib:-in-tree (warm 3)}" | tee -a $out/ab.txt
  CHITU_HIP_LIB=$L timeout 200 python tools/prefill_bench.py 8 512 2048 8192 2>/dev/null | grep prompt_tokens | tee -a $out/ab.txt
done; done
cd /tmp && export TMPDIR=/tmp
for lib in build_probe/lib_mwarm0.so build_probe/lib_mwarm2.so "" build_probe/lib_mwarm5.so; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  rm -rf /tmp/pq; CHITU_HIP_LIB=$L timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pq -o t -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py 8 2048 > /tmp/pq.log 2>&1
  echo "== kernel trace at 2048 tokens, ${lib:-in-tree (warm 3)}" | tee -a $out/ab.txt
  timeout 60 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pq/t_results.db --last-fraction 0.3 | grep -E "moe_gemm_tiled" | cut -c1-150 | tee -a $out/ab.txt
done
