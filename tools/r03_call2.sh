#!/bin/bash
# round-3 GPU call 2: the new bf16-activation MoE modes + the reference's own model on the HIP operator surface.
# The drop-in test needs a copy of the reference's `chitu/` package on the GPU box: stage it (git-ignored, removed after the
# call) with   mkdir -p gpurun_stage && cp -r /root/reference/chitu gpurun_stage/
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call2
mkdir -p $out
cd $GRAFT_REPO_ROOT
CHITU_REFERENCE_DIR=$GRAFT_REPO_ROOT/gpurun_stage timeout 900 python -m pytest tests/test_gpu_reference_dropin.py -x -q -s > $out/dropin.txt 2>&1
echo "dropin rc=$?" >> $out/dropin.txt
timeout 900 python -m pytest tests/test_gpu_moe.py -x -q > $out/moe.txt 2>&1
echo "moe rc=$?" >> $out/moe.txt
tail -30 $out/dropin.txt; tail -15 $out/moe.txt
