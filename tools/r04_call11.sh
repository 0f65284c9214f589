#!/bin/bash
# round 4, call 11: the ring-depth heuristic in the unfused bf16 GEMM (grids above one workgroup per CU), Llama-3-8B bs 4 / 16
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call11
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 python tools/llama_ab.py --bs 4,16 --reps 2 --steps 40 --opt bf16_gemm_deep=1 > $out/llama_ab.txt 2>&1; echo "rc=$?" >> $out/llama_ab.txt
grep -v amdgpu.ids $out/llama_ab.txt | grep -v "prologue" | tail -12 | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_llama.py tests/test_gpu_mixtral.py -m gpu -q --timeout 300 > $out/tests.txt 2>&1; echo "rc=$?" >> $out/tests.txt
tail -3 $out/tests.txt | cut -c1-300
timeout 200 python tools/run_extra.py mixtral 32 2>/dev/null | tail -1 | cut -c1-700
