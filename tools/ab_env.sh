#!/bin/bash
# A/B an environment knob on the SAME GPU box: tools/ab_env.sh VAR=off_value [bench.py args...]
#   -> ms/step (bs16, bs1) with the knob set, then unset, twice
knob=$1; shift
for rep in 1 2; do
  for mode in "$knob" ""; do
    env $mode python bench.py --no-cpu-baseline --no-roofline --no-llama "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('${mode:-default}', d['ms_per_step'], d.get('bs1',{}).get('ms_per_step'))"
  done
done
