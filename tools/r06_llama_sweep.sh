#!/bin/bash
# Launch-variant sweep on the Llama-3-8B and V2-Lite extras (after the Mixtral finding that a K-split heuristic was off by 12 %).
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_sweep; mkdir -p $out
line() { python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k: (v['ms_per_step'], v['roofline_frac']) for k, v in d.items() if k.startswith('bs')})"; }
for opt in "" "bf16_gemm_wk=2" "bf16_gemm_wk=4" "bf16_gemm_wk=8" "bf16_silu_wk=2" "bf16_silu_wk=4" "bf16_silu_wk=8" "bf16_gemm_deep=0" "bf16_gemm_deep=1" ""; do
  echo "== llama CHITU_DEBUG_OPTIONS=$opt" | tee -a $out/llama.txt
  CHITU_DEBUG_OPTIONS=$opt timeout 200 python tools/run_extra.py llama 16 2>/dev/null | tail -1 | line | tee -a $out/llama.txt
done
for opt in "" "moe_gemm1_wk=2" "moe_gemm1_wk=4" "moe_gemm1_wk=8" "moe_gemm2_cfg=1" "moe_gemm2_cfg=2" "moe_gemm2_cfg=3" "fp8_gemm_wk=4" "fp8_gemm_wk=8" ""; do
  echo "== v2lite CHITU_DEBUG_OPTIONS=$opt" | tee -a $out/v2lite.txt
  CHITU_DEBUG_OPTIONS=$opt timeout 200 python tools/run_extra.py v2lite 16 2>/dev/null | tail -1 | line | tee -a $out/v2lite.txt
done
