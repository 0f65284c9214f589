#!/bin/bash
# round 4, call 10: ring depth of the fused qkv projection when the grid exceeds the CU count (Llama-3-8B), same-box A/B;
# bit-identity of the launch variants; the bench line's second roofline leg
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call10
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 python tools/llama_ab.py --bs 1,2 --reps 2 --steps 40 --opt bf16_gemm_deep=1 > $out/llama_ab.txt 2>&1; echo "rc=$?" >> $out/llama_ab.txt
grep -v amdgpu.ids $out/llama_ab.txt | tail -20 | cut -c1-300
timeout 400 python -m pytest tests/test_gpu_llama.py -m gpu -q --timeout 300 > $out/tests.txt 2>&1; echo "rc=$?" >> $out/tests.txt
tail -4 $out/tests.txt | cut -c1-300
timeout 300 python bench.py --no-llama --no-bs1 --no-cpu-baseline --steps 32 2>$out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], json.dumps(d['roofline_kernels'])[:1200])"
grep -v amdgpu.ids $out/bench_err.txt | head -5
