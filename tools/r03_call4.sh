#!/bin/bash
# round-3 GPU call 4: hand-off probe with one poller per XCC; tile-major activation addressing (timing-only probe builds)
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call4
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 180 tools/probe_handoff.bin 2>&1 | grep -E "^B|part B" > $out/probe_handoff_B.txt
for rep in 1 2; do
for v in "" gemm16; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/build_probe/lib_$v.so
  echo "== ${v:-product}" >> $out/dense.txt
  CHITU_HIP_LIB=$L timeout 300 python tools/bench_kernels.py --only dense --bs 16 2>&1 | grep -v amdgpu.ids >> $out/dense.txt
done
for v in "" moe1; do
  L=""; [ -n "$v" ] && L=$GRAFT_REPO_ROOT/build_probe/lib_$v.so
  echo "== ${v:-product}" >> $out/moe.txt
  CHITU_HIP_LIB=$L timeout 300 python tools/bench_kernels.py --only moe --bs 16 32 2>&1 | grep -E "gemm1_silu|gemm2_quant" >> $out/moe.txt
done
done
cat $out/probe_handoff_B.txt $out/dense.txt $out/moe.txt
