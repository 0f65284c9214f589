#!/bin/bash
# Round 6: the whole library built with -fno-slp-vectorize (no compiler-formed v_pk_*_f32) against the in-tree build, same box:
# (1) the race hunt (two rank processes on one GPU, allocator poisoned: the in-tree build fails a few percent of its steps in the in-place RoPE
#     of absorb_bmm_kernel, a v_pk_mul_f32 / v_pk_add_f32 op_sel + neg sequence), (2) N = 2 shared-GPU bench (graph == eager check at bs 16 / 32),
# (3) step time at bs 16 / 1 / 32 and the prefill section, twice each.
cd $GRAFT_REPO_ROOT
out=$GRAFT_REPO_ROOT/gpurun_out/r06_noslp; mkdir -p $out
NOSLP=$GRAFT_REPO_ROOT/build_probe/lib_noslp.so
for lib in "" $NOSLP; do
  tag=$([ -n "$lib" ] && echo noslp || echo intree)
  for i in 1 2 3 4; do
    CHITU_HIP_LIB=$lib HUNT_POISON=1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2970$i tools/r06_race_hunt.py 32 12 30 0 2>&1 | grep "repetitions differ" >> $out/hunt_$tag.txt
  done
  echo "== hunt $tag"; cat $out/hunt_$tag.txt
  CHITU_HIP_LIB=$lib timeout 500 python bench.py --gpus 2 --layers 12 --steps 8 --warmup 2 --no-llama --no-cpu-baseline --no-roofline > $out/n2_$tag.json 2> $out/n2_$tag.err; echo "n2 $tag rc=$?"
  grep -h "failed its replay check\|does not reproduce" $out/n2_$tag.err | sort | uniq -c | cut -c1-200
done
for rep in 1 2; do for lib in "" $NOSLP; do
  CHITU_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-roofline --no-llama | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-in-tree}', d['ms_per_step'], d.get('bs1',{}).get('ms_per_step'), d.get('bs32',{}).get('ms_per_step'))"
  CHITU_HIP_LIB=$lib timeout 300 python tools/prefill_bench.py 8 2048 2>/dev/null | grep prompt_tokens
done; done 2>&1 | grep -v amdgpu.ids | tee $out/ab_step.txt
