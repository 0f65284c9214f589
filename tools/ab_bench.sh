#!/bin/bash
# A/B two builds of libchitu_hip.so on the SAME GPU box (boxes differ by ~15% on latency-bound kernels):
#   tools/ab_bench.sh <baseline.so> [bench.py args...]   -> ms/step (bs16, bs1) for baseline then current, twice
base=$1; shift
for rep in 1 2; do
  for lib in "$base" ""; do
    CHITU_HIP_LIB=$lib python bench.py --no-cpu-baseline --no-roofline "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${lib:-current}'.split('/')[-1], d['ms_per_step'], d.get('bs1',{}).get('ms_per_step'))"
  done
done
