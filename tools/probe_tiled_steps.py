#!/usr/bin/env python3
"""Shader-clock stamps inside fp8_gemm_tiled_kernel (probe build of that one file):
    tools/build_variant.sh tiledprobe fp8_gemm_tiled.hip -DCHITU_PROBE
    CHITU_HIP_LIB=build_probe/lib_tiledprobe.so python tools/probe_tiled_steps.py [tokens] [N] [K] [tm]
prints, for K steps 8..13 of workgroup 0, the cycles between: step top -> own DMA pieces landed -> barrier passed -> next stage
requested -> block multiplied (-> next step's top).
    tools/build_variant.sh moeprobe moe_tiled.hip -DCHITU_PROBE
    CHITU_HIP_LIB=build_probe/lib_moeprobe.so python tools/probe_tiled_steps.py moe [tokens] [experts] [topk]
the same stamps inside moe_gemm_tiled_kernel<SILU> (GEMM1 of a prefill-sized fused_experts call at R1's per-rank expert shapes; GEMM2's
eight steps do not reach step 8)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chitu_amd import _lib, ops

def moe_mode():
    from chitu_amd import fused_moe
    T, E, topk = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((2, 2048), (3, 256), (4, 8)))
    K, I = 7168, 256
    gd = torch.Generator(device="cuda").manual_seed(7)
    x = (torch.randn(T, K, device="cuda", generator=gd) * 0.5).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * I, K, device="cuda", generator=gd) * 0.5).to(torch.float8_e4m3fn)
    w2 = (torch.randn(E, K, I, device="cuda", generator=gd) * 0.5).to(torch.float8_e4m3fn)
    w1s = torch.rand(E, 2 * I // 128, K // 128, device="cuda", generator=gd) * 0.02 + 0.01
    w2s = torch.rand(E, K // 128, I // 128, device="cuda", generator=gd) * 0.02 + 0.01
    ids = torch.stack([torch.randperm(E, device="cuda", generator=gd)[:topk] for _ in range(T)])
    wts = torch.rand(T, topk, device="cuda", generator=gd).to(torch.bfloat16)
    for _ in range(3):
        fused_moe.fused_experts(x.clone(), w1, w2, wts, ids, inplace=False, use_fp8_w8a8=True, w1_scale=w1s, w2_scale=w2s, block_shape=[128, 128])
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    assert _lib.lib().chitu_hip_probe_read_moe_tiled(buf) == 0
    print(f"moe GEMM1: tokens {T} experts {E} topk {topk}: cycles per phase, steps 8..13 of workgroup 0 (wait DMA | barrier | issue next | LDS reads + MFMA + fold | total)")
    for s in range(6):
        m = [buf[s * 5 + i] for i in range(5)]
        nxt = buf[(s + 1) * 5] if s < 5 else None
        d = [m[i + 1] - m[i] for i in range(4)]
        print(f"  step {8 + s}: {d[0]:6d} | {d[1]:6d} | {d[2]:6d} | {d[3]:6d} | " + (f"{nxt - m[0]:6d}" if nxt else "     -"))


if len(sys.argv) > 1 and sys.argv[1] == "moe":
    moe_mode()
    sys.exit(0)
T, N, K = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 2048), (2, 2112), (3, 7168)))
tm = int(sys.argv[4]) if len(sys.argv) > 4 else -1
gd = torch.Generator(device="cuda").manual_seed(5)
x = torch.randn(T, K, device="cuda", generator=gd).to(torch.bfloat16)
xq, xs = ops.act_quant_deepseek_v3(x)
w = (torch.randn(N, K, device="cuda", generator=gd) * 0.5).to(torch.float8_e4m3fn)
ws = torch.rand((N + 127) // 128, (K + 127) // 128, device="cuda", generator=gd) * 0.02 + 0.01
with _lib.debug_option("fp8_tiled_tm", tm):
    for _ in range(4):
        ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
assert _lib.lib().chitu_hip_probe_read_fp8_gemm_tiled(buf) == 0
print(f"tokens {T} N {N} K {K} tm {tm}: cycles per phase, K steps 8..13 of workgroup 0 (wait DMA | barrier | issue next | LDS reads + MFMA + fold | total)")
for s in range(6):
    m = [buf[s * 5 + i] for i in range(5)]
    nxt = buf[(s + 1) * 5] if s < 5 else None
    d = [m[i + 1] - m[i] for i in range(4)]
    print(f"  step {8 + s}: {d[0]:6d} | {d[1]:6d} | {d[2]:6d} | {d[3]:6d} | " + (f"{nxt - m[0]:6d}" if nxt else "     -"))
