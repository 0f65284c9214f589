#!/usr/bin/env python3
"""Do the small kernels of a layer get slower when the expert GEMM in front of them streams more
distinct experts (= touches more pages)?  Graph = L x [gemm1(D distinct experts) + 4 rmsnorm(add,quant)]
vs L x [gemm1(D)]; the difference / (4L) is the in-context cost of one small kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chitu_amd import _lib, fused_moe, ops
from chitu_amd._lib import i32, i64, ptr, stream_ptr

torch.cuda.set_device(0)
dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(0)
FP8 = torch.float8_e4m3fn
L, E, K, I, topk, bs = 8, 257, 7168, 256, 9, 16
lib = _lib.lib()
w1 = [torch.randint(0, 100, (E, 2 * I, K), device=dev, dtype=torch.uint8).view(FP8) for _ in range(L)]
w1s = [torch.rand(E, 4, 56, device=dev) * 0.02 + 0.01 for _ in range(L)]
x = torch.randn(bs, K, device=dev, dtype=torch.bfloat16)
xq, xs = fused_moe.per_token_group_quant_fp8(x, 128)
numel = bs * topk
c1 = torch.empty(numel, 2 * I, dtype=torch.bfloat16, device=dev)
nw = [torch.ones(K, device=dev, dtype=torch.bfloat16) for _ in range(4 * L)]
res = [torch.randn(bs, K, device=dev, dtype=torch.bfloat16) for _ in range(4)]

def replay_ms(fn):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[3]

for D in (16, 40, 60, 70, 80, 90, 100, 110, 128):
    # exactly D distinct routed experts (spread over the table) + the shared one
    pool = torch.randperm(E - 1, device=dev, generator=gen)[:D]
    flat = torch.cat([pool, pool[torch.randint(0, D, (bs * 8 - D,), device=dev, generator=gen)]]) if D <= bs * 8 else pool[: bs * 8]
    ids = torch.cat([flat.view(bs, 8), torch.full((bs, 1), E - 1, device=dev)], 1).contiguous()
    sid, eid, npost = fused_moe.moe_align_block_size(ids, 16, E)
    mmb = min(eid.numel(), numel)
    def gemm(l):
        rc = lib.chitu_hip_moe_gemm1_fp8(ptr(xq), ptr(xs), ptr(w1[l]), ptr(w1s[l]), ptr(sid), ptr(eid), ptr(npost), ptr(c1),
                                         i64(numel), i32(topk), i64(2 * I), i64(K), i64(mmb), stream_ptr())
        assert rc == 0
    def only_gemm():
        for l in range(L): gemm(l)
    def with_small():
        for l in range(L):
            gemm(l)
            for k in range(4):
                ops.rms_norm(res[k], nw[4 * l + k], 1e-6, out_bf16=False, quant="act", add=res[(k + 1) % 4])
    a, b = replay_ms(only_gemm), replay_ms(with_small)
    print(f"D={int(ids.unique().numel()):4d}  gemm1 {a/L*1e3:7.2f} us   rmsnorm in context {(b-a)/(4*L)*1e3:6.2f} us", flush=True)
