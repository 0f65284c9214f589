#!/bin/bash
# round 4, call 7: split-phase world 4 / 8, tightened MoE parity, V2-Lite and EP-rank extras
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call7
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_moe.py tests/test_gpu_deepseek.py tests/test_gpu_moe_align.py -m gpu -q --timeout 300 -k "reference_fixture or vs_oracle or router_gemm_prologue or gate or align" > $out/tests_a.txt 2>&1; echo "rc=$?" >> $out/tests_a.txt
tail -6 $out/tests_a.txt | cut -c1-300
timeout 500 python -m pytest tests/test_gpu_xgmi.py -m gpu -q --timeout 450 -k "four_rank_processes" > $out/tests_b.txt 2>&1; echo "rc=$?" >> $out/tests_b.txt
tail -12 $out/tests_b.txt | cut -c1-600
(timeout 700 python tools/xgmi_world8.py --split-phase 8 3 600; echo "rc=$?") > $out/world8.txt 2>&1
(timeout 500 python tools/xgmi_world8.py --split-phase 4 3 400; echo "rc=$?") > $out/world4.txt 2>&1
grep -v "amdgpu.ids\|Gloo" $out/world8.txt | tail -8 | cut -c1-400
grep -v "amdgpu.ids\|Gloo" $out/world4.txt | tail -8 | cut -c1-400
for e in "X=1" "CHITU_MOE_TWO_LAUNCH_MAX_I=2048"; do
  echo "== v2lite $e" >> $out/extras.txt
  env $e timeout 300 python tools/run_extra.py v2lite 32 2>/dev/null | tail -1 >> $out/extras.txt
done
echo "== ep8" >> $out/extras.txt
timeout 300 python tools/run_extra.py ep8 32 2>/dev/null | tail -1 >> $out/extras.txt
cat $out/extras.txt | cut -c1-900
