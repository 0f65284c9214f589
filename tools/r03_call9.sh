#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call9
mkdir -p $out
cd $GRAFT_REPO_ROOT
for opt in "" "moe_gemm2_cfg=22" "moe_gemm2_cfg=24" "moe_gemm2_cfg=28" "moe_gemm2_cfg=42" "moe_gemm1_d=2" "moe_gemm1_d=4" "moe_gemm1_wk=2"; do
  O=""; [ -n "$opt" ] && O="--opt $opt"
  echo "== ${opt:-default}" >> $out/moe_sweep.txt
  timeout 300 python tools/bench_kernels.py --only moe --bs 16 32 $O 2>&1 | grep -E "gemm1_silu|gemm2_quant" >> $out/moe_sweep.txt
done
cat $out/moe_sweep.txt
