#!/bin/bash
# round-3 GPU call 3: what are the latency-bound kernels made of?  Phase-masked probe builds (tools/build_variant.sh) of the
# dense fp8 GEMM and the MLA decode kernel, each timed by tools/bench_kernels.py (graph-replayed, HBM-cold weights) on ONE box.
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call3
mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in "" gemm1 gemm2 gemm4 gemm5 gemm8 gemm13; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/build_probe/lib_$v.so
  echo "== ${v:-product}" >> $out/dense.txt
  CHITU_HIP_LIB=$L timeout 300 python tools/bench_kernels.py --only dense --bs 1 16 2>&1 | grep -v amdgpu.ids >> $out/dense.txt
done
for v in "" mla1 mla2 mla8 mla16 mla27; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/build_probe/lib_$v.so
  echo "== ${v:-product}" >> $out/mla.txt
  CHITU_HIP_LIB=$L timeout 300 python tools/bench_kernels.py --only mla --bs 1 16 2>&1 | grep -v amdgpu.ids >> $out/mla.txt
done
timeout 300 python tools/bench_kernels.py --only small --bs 1 16 2>&1 | grep -v amdgpu.ids > $out/small.txt
cat $out/dense.txt $out/mla.txt $out/small.txt
