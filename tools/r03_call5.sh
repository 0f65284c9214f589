#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r03_call5
mkdir -p $out
cd $GRAFT_REPO_ROOT
for bs in 16 1; do
CHITU_HIP_LIB=$GRAFT_REPO_ROOT/build_probe/libchitu_hip_probe.so timeout 300 python tools/probe_phases.py $bs 2>&1 | grep -v amdgpu.ids > $out/phases_bs$bs.txt
done
cat $out/phases_bs16.txt $out/phases_bs1.txt
