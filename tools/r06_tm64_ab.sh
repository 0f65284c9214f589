#!/bin/bash
# the tiled fp8 GEMM with 64- against 128-token tiles (launch-variant option fp8_tiled_tm) on the R1 prefill shapes, same box, same library:
#   gpurun -- bash tools/r06_tm64_ab.sh   -> [us, TFLOP/s] per shape@tokens
cd $GRAFT_REPO_ROOT
# (64-token tiles: three ring stages in-tree, two in build_probe/lib_ns64_2.so = tools/build_variant.sh ns64_2 fp8_gemm_tiled.hip -DCHITU_TILED_NS64=2)
for rep in 1 2; do for cfg in 128: 64: 64:build_probe/lib_ns64_2.so; do
  tm=${cfg%%:*}; lib=${cfg#*:}; L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  echo -n "tm=$tm ${lib:-in-tree} "
  CHITU_HIP_LIB=$L CHITU_DEBUG_OPTIONS=fp8_tiled_tm=$tm timeout 200 python - <<'PY' 2>/dev/null
import json, os, torch, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from chitu_amd import ops, _lib
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
