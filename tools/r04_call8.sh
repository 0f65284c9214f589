#!/bin/bash
# round 4, call 8: phase-ordered MLA prefill kernel (parity + same-box A/B), split-phase world 4 / 8, MoE parity bar
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call8
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_mla.py tests/test_gpu_moe.py tests/test_gpu_deepseek.py -m gpu -q --timeout 300 -k "prefill or vs_oracle or reference_fixture" > $out/tests_a.txt 2>&1; echo "rc=$?" >> $out/tests_a.txt
tail -6 $out/tests_a.txt | cut -c1-300
for rep in 1 2; do
  for lib in build_probe/lib_prefill_r03.so chitu_amd/libchitu_hip.so; do
    echo "== $lib" >> $out/prefill_ab.txt
    CHITU_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python tools/prefill_bench.py 8 2>/dev/null | tail -4 >> $out/prefill_ab.txt
  done
done
cat $out/prefill_ab.txt | cut -c1-300
timeout 500 python -m pytest tests/test_gpu_xgmi.py -m gpu -q --timeout 450 -k "four_rank_processes" > $out/tests_b.txt 2>&1; echo "rc=$?" >> $out/tests_b.txt
grep -v "amdgpu.ids\|Gloo\|socket.cpp" $out/tests_b.txt | tail -8 | cut -c1-700
(timeout 700 python tools/xgmi_world8.py --split-phase 8 3 600; echo "rc=$?") > $out/world8.txt 2>&1
(timeout 500 python tools/xgmi_world8.py --split-phase 4 3 400; echo "rc=$?") > $out/world4.txt 2>&1
grep -v "amdgpu.ids\|Gloo\|socket.cpp" $out/world8.txt | tail -6 | cut -c1-700
grep -v "amdgpu.ids\|Gloo\|socket.cpp" $out/world4.txt | tail -6 | cut -c1-700
