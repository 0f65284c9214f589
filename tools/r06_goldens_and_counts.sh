#!/bin/bash
# round 6: new hardware goldens (GQA via the reference's pure-torch attention, soft-fp8 MoE branch), the MoE outlier census,
# the fused-tail A/B after the departure fix.  Needs gpurun_stage/reference.
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_goldens; mkdir -p $out
export CHITU_REFERENCE_DIR=$PWD/gpurun_stage/reference PYTHONDONTWRITEBYTECODE=1
HW_GOLDEN_PROFILE=$out/reference_on_mi355x_r06.txt timeout 600 python tests/golden/gen_hw_golden.py gqa soft_fp8_moe > $out/gen.txt 2>&1; echo "gen rc=$?" >> $out/gen.txt
cp tests/golden/hw_gqa.npz tests/golden/hw_soft_fp8_moe.npz $out/ 2>/dev/null
tail -15 $out/gen.txt
timeout 300 python tools/r06_moe_outliers.py > $out/moe_outliers.txt 2>&1; cat $out/moe_outliers.txt | grep -v amdgpu.ids
timeout 600 python -m pytest tests/test_gpu_gqa.py tests/test_gpu_moe.py -q -k "mi355x" > $out/tests.txt 2>&1; tail -5 $out/tests.txt
bash tools/ab_env.sh CHITU_MLA_FUSED_TAIL=0 > $out/ab_step_fused_tail.txt 2>&1; grep -v amdgpu.ids $out/ab_step_fused_tail.txt
for bs in 16 1; do
  bash tools/ab_env_kernel_time.sh "mla_decode|mla_merge" $bs "CHITU_MLA_FUSED_TAIL=0" "" > $out/ab_kernel_fused_tail_bs$bs.txt 2>&1; cat $out/ab_kernel_fused_tail_bs$bs.txt
done
