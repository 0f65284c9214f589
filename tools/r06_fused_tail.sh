#!/bin/bash
# Round 6 item 2: the fused MLA decode tail -- parity tests, then same-box A/B (step time and kernel time) against the two launches.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_fused_tail; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_mla.py tests/test_gpu_deepseek.py -x -q > $out/tests.txt 2>&1; echo "rc=$?" >> $out/tests.txt
tail -5 $out/tests.txt
bash tools/ab_env.sh CHITU_MLA_FUSED_TAIL=0 > $out/ab_step.txt 2>&1; cat $out/ab_step.txt
for bs in 16 1; do
  bash tools/ab_env_kernel_time.sh "mla_decode|mla_merge" $bs "CHITU_MLA_FUSED_TAIL=0" "" > $out/ab_kernel_bs$bs.txt 2>&1; cat $out/ab_kernel_bs$bs.txt
done
