#!/bin/bash
# the prefill expert GEMMs with the step's fragment reads hoisted out of the per-sub-tile skip (in-tree) against the kernel of the earlier commits
# (build_probe/lib_moeold.so), same box: the expert GEMMs' kernel time in 8 layers of the R1 rank shard at 2048 / 8192 / 1024 tokens
cd /tmp && export TMPDIR=/tmp
for T in 2048 8192 1024; do for lib in build_probe/lib_moeold.so "" build_probe/lib_moeold.so ""; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  rm -rf /tmp/pq; CHITU_HIP_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pq -o t -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py 8 $T > /tmp/pq.log 2>&1
  echo "== $T tokens, ${lib:-in-tree}: $(grep prompt_tokens /tmp/pq.log | tail -1 | cut -c1-110)"
  timeout 60 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pq/t_results.db --last-fraction 0.3 | grep -E "moe_gemm_tiled" | cut -c1-150
done; done
