#!/bin/bash
# Round 6 item 5: 128-slot tiles (sub-tile skipping, two stages) for the prefill expert GEMMs against the 64-slot form, same box, same library:
# tools/prefill_bench.py under CHITU_MOE_TILED_BLOCK_M=64 / 128, twice; then one kernel trace each at 2048 tokens.  (Parity: tests/test_gpu_moe.py,
# test_gpu_production_shapes.py, test_gpu_deepseek.py -k "tiled or prefill or production or mi355x": 23 passed under either block size.)
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_moe128; mkdir -p $out; rm -f $out/ab.txt
for rep in 1 2; do for bm in 64 128; do
  echo "== block_m $bm" | tee -a $out/ab.txt
  CHITU_MOE_TILED_BLOCK_M=$bm timeout 200 python tools/prefill_bench.py 8 512 2048 8192 2>/dev/null | grep prompt_tokens | tee -a $out/ab.txt
done; done
cd /tmp && export TMPDIR=/tmp
for bm in 64 128; do
  rm -rf /tmp/pq; CHITU_MOE_TILED_BLOCK_M=$bm timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/pq -o t -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py 8 2048 > /tmp/pq.log 2>&1
  echo "== kernel trace, block_m $bm" | tee -a $out/ab.txt
  timeout 60 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py /tmp/pq/t_results.db --last-fraction 0.3 | grep -E "moe_gemm_tiled|moe_align|rmsnorm_add_kernel<1, 16>" | cut -c1-150 | tee -a $out/ab.txt
done
