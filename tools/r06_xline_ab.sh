#!/bin/bash
# same-box A/B of the skinny bf16 GEMM's activation loads: whole lines + lane swap (in-tree) against two half-line loads per token
# (build_probe/lib_xline0.so = tools/build_variant.sh xline0 gate.hip -DCHITU_BF16_XLINE=0):  gpurun -- bash tools/r06_xline_ab.sh
# (measured equal, profiles/r06_ab_bf16_xline.txt; the knob left the kernel with the experiment)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in build_probe/lib_xline0.so ""; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  echo "== ${lib:-in-tree}"
  CHITU_HIP_LIB=$L timeout 200 python - <<'PY' 2>/dev/null
import json, os, torch, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from chitu_amd import ops
def time_us(fn, n=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
gd = torch.Generator(device="cuda").manual_seed(5)
row = {}
ws = {name: (torch.randn(N, K, device="cuda", generator=gd) * 0.05).to(torch.bfloat16) for name, (N, K) in
      {"qkv": (6144, 4096), "wo": (4096, 4096), "w2": (4096, 14336), "router": (256, 7168), "head": (128256, 4096)}.items()}
for M in (1, 4, 16, 32):
    for name, w in ws.items():
        x = torch.randn(M, w.shape[1], device="cuda", generator=gd).to(torch.bfloat16)
        us = time_us(lambda: ops.bf16_linear(x, w))
        row[f"{name}@{M}"] = [round(us, 1), round(w.numel() * 2 / us * 1e-6, 2)]
print(json.dumps(row))
PY
  CHITU_HIP_LIB=$L CHITU_BENCH_EXTRA_BATCHES=1,16 timeout 300 python tools/run_extra.py llama 16 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(v['ms_per_step'],v['roofline_frac']) for k,v in d.items() if k.startswith('bs')})"
done; done
