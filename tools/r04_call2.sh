#!/bin/bash
# round 4, call 2: stale-kernel-argument hypothesis (decoy graphs with other weights destroyed before the instance under test)
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call2
mkdir -p $out
cd $GRAFT_REPO_ROOT
run() { echo "=== $*" >> $out/repro2.txt; env "$@" timeout 300 python tools/graph_repro2.py $ARGS >> $out/repro2.txt 2>&1; echo "rc=$?" >> $out/repro2.txt; }
ARGS="--trials 6 --decoys 4" run X=1
ARGS="--trials 6 --decoys 4" run GPU_MAX_HW_QUEUES=8
ARGS="--trials 4 --decoys 12 --empty-cache 0" run GPU_MAX_HW_QUEUES=8
ARGS="--trials 4 --decoys 12 --keep 3" run GPU_MAX_HW_QUEUES=8
ARGS="--trials 6 --decoys 4" run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
ARGS="--trials 6 --decoys 4" run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
ARGS="--trials 6 --decoys 4" run HIP_FORCE_DEV_KERNARG=0
grep -v "amdgpu.ids" $out/repro2.txt | tail -120
