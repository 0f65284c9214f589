#!/bin/bash
# round-3 GPU call 2: the new bf16-activation MoE modes + the reference's own model on the HIP operator surface
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call2
mkdir -p $out
cd $GRAFT_REPO_ROOT
true
true
timeout 900 python -m pytest tests/test_gpu_moe.py -x -q > $out/moe.txt 2>&1
echo "moe rc=$?" >> $out/moe.txt
tail -15 $out/moe.txt
