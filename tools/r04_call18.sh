#!/bin/bash
# round 4, call 18: the one-query-token-per-wave MLA prefill kernel -- parity (vs the exact kernel, the oracle, the fixture) and a same-box A/B
out=$GRAFT_REPO_ROOT/gpurun_out/r04_call18
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 60 python -m pytest tests/test_gpu_mla.py -m gpu -q --timeout 50 -k "one_token_per_wave" > $out/tests.txt 2>&1; echo "rc=$?" >> $out/tests.txt
grep -v amdgpu.ids $out/tests.txt | tail -14 | cut -c1-500
for mode in kernel tiled; do
  echo "== CHITU_MLA_PREFILL=$mode" >> $out/prefill_ab.txt
  CHITU_MLA_PREFILL=$mode timeout 45 python tools/prefill_bench.py 8 2>/dev/null | grep prompt_tokens >> $out/prefill_ab.txt
done
cat $out/prefill_ab.txt
