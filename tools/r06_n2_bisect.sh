#!/bin/bash
# Is the N = 2 shared-GPU graph != eager mismatch (bs 32, profiles/r06...) new in round 6?  Same command on (a) the round-5 final tree
# (gpurun_stage/r05tree), (b) this tree without the enable_xgmi pre-flight, (c) this tree.  Only the capture messages and the exit code matter.
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_n2_bisect; mkdir -p $out
run() {  # name dir env...
  name=$1; dir=$2; shift 2
  ( cd $dir && env "$@" timeout 500 python bench.py --gpus 2 --layers 12 --steps 8 --warmup 2 --no-llama --no-cpu-baseline --no-roofline > $out/$name.json 2> $out/$name.err; echo "$name rc=$?" )
  grep -h "failed its replay check\|does not reproduce" $out/$name.err | sort | uniq -c | cut -c1-260
}
run r05tree $GRAFT_REPO_ROOT/gpurun_stage/r05tree A=1
run r06_no_preflight $GRAFT_REPO_ROOT CHITU_XGMI_PREFLIGHT=0
run r06 $GRAFT_REPO_ROOT A=1
run r06_rccl_path $GRAFT_REPO_ROOT CHITU_ALLREDUCE=rccl
