#!/bin/bash
# the tiled fp8 GEMM as shipped at the end of round 6 (64-token tiles for small grids, one-barrier step order) against the kernel of the
# round's earlier commits (build_probe/lib_warm0.so: 128-token tiles only, request -> multiply -> wait -> barrier), same box:
#   gpurun -- bash tools/r06_tiled_final_ab.sh   -> [us, TFLOP/s] per shape@tokens(128-token tiles)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for cfg in ${CFGS:-"build_probe/lib_warm0.so:" ":" ":fp8_tiled_tm=128"}; do
  lib=${cfg%%:*}; opt=${cfg#*:}; L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  echo -n "${lib:-in-tree} ${opt:-heuristic} "
  CHITU_HIP_LIB=$L CHITU_DEBUG_OPTIONS=$opt timeout 200 python - <<'PY' 2>/dev/null
import json, os, torch, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from chitu_amd import ops, _lib
if "lib_warm0" not in os.environ.get("CHITU_HIP_LIB", ""):
    _lib.apply_debug_options_from_env()
def time_us(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
gd = torch.Generator(device="cuda").manual_seed(5)
row = {}
for T in (256, 512, 1024, 2048, 4096, 8192):
    for name, (N, K) in {"wqkv_a": (2112, 7168), "wq_b": (3072, 1536), "wo": (7168, 2048), "dense_w1w3": (4608, 7168), "dense_w2": (7168, 2304)}.items():
        x = torch.randn(T, K, device="cuda", generator=gd).to(torch.bfloat16)
        xq, xs = ops.act_quant_deepseek_v3(x)
        w = (torch.randn(N, K, device="cuda", generator=gd) * 0.5).to(torch.float8_e4m3fn)
        ws = torch.rand((N + 127) // 128, (K + 127) // 128, device="cuda", generator=gd) * 0.02 + 0.01
        us = time_us(lambda: ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.bfloat16))
        tiles = ((T + 127) // 128) * ((N + 127) // 128)
        row[f"{name}@{T}({tiles})"] = [round(us, 1), round(2.0 * T * N * K / us * 1e-6)]
print(json.dumps(row))
PY
done; done
