#!/bin/bash
# same-box A/B of the tiled fp8 GEMM's L2 warm-up distance (CHITU_TILED_WARM = 0 / 2 / 4 (in-tree) / 6 / 8): build_probe/lib_warm<N>.so from
# tools/build_variant.sh warm<N> fp8_gemm_tiled.hip -DCHITU_TILED_WARM=<N>;  gpurun -- bash tools/r06_warm_ab.sh
# (the knob left the kernel with the experiment -- measured slower, profiles/r06_ab_tiled_l2_warmup.txt; kept as the record of how it was timed)
cd $GRAFT_REPO_ROOT
for lib in build_probe/lib_warm0.so build_probe/lib_warm2.so "" build_probe/lib_warm6.so build_probe/lib_warm8.so build_probe/lib_warm0.so ""; do
  L=""; [ -n "$lib" ] && L=$GRAFT_REPO_ROOT/$lib
  CHITU_HIP_LIB=$L timeout 120 python - <<'PY' 2>/dev/null
import json, os, torch, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from chitu_amd import ops
def time_us(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
gd = torch.Generator(device="cuda").manual_seed(5)
row = {}
for T in (512, 2048, 8192):
    for name, (N, K) in {"wqkv_a": (2112, 7168), "wq_b": (3072, 1536), "wo": (7168, 2048), "dense_w1w3": (4608, 7168), "dense_w2": (7168, 2304)}.items():
        x = torch.randn(T, K, device="cuda", generator=gd).to(torch.bfloat16)
        xq, xs = ops.act_quant_deepseek_v3(x)
        w = (torch.randn(N, K, device="cuda", generator=gd) * 0.5).to(torch.float8_e4m3fn)
        ws = torch.rand((N + 127) // 128, (K + 127) // 128, device="cuda", generator=gd) * 0.02 + 0.01
        us = time_us(lambda: ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.bfloat16))
        row[f"{name}@{T}"] = [round(us, 1), round(2.0 * T * N * K / us * 1e-6)]
print(os.path.basename(os.environ.get("CHITU_HIP_LIB") or "in-tree(warm4)"), json.dumps(row))
PY
done
