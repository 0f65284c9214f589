#!/usr/bin/env python3
"""Stand-alone timing of the hot kernels at DeepSeek-R1 TP=8 per-rank shapes (HIP events, HBM-cold
weights: each launch uses a different layer's buffers, > 256 MB apart).  Tuning aid, not the bench."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chitu_amd import _lib, fused_moe, ops
from chitu_amd._lib import i32, i64, f32, ptr, stream_ptr

ap = argparse.ArgumentParser()
ap.add_argument("--bs", type=int, nargs="+", default=[1, 16])
ap.add_argument("--layers", type=int, default=8)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--only", type=str, default="")
ap.add_argument("--opt", type=str, default="", help="launch-variant overrides, e.g. fp8_gemm_wk=8,fp8_gemm_deep=0")
a = ap.parse_args()
for kv in filter(None, a.opt.split(",")):
    k, v = kv.split("=")
    _lib.check(_lib.lib().chitu_hip_debug_option(i32(_lib.DEBUG_OPTIONS[k]), i32(int(v))), "debug_option")
    print(f"# override {k} = {v}")
torch.cuda.set_device(0)
dev = "cuda"
gen = torch.Generator(device=dev).manual_seed(0)
FP8 = torch.float8_e4m3fn


def rfp8(*shape):
    t = torch.empty(*shape, dtype=FP8, device=dev)
    flat = t.view(-1)
    for i in range(0, flat.numel(), 1 << 26):
        n = min(1 << 26, flat.numel() - i)
        flat[i:i + n].copy_((torch.randn(n, device=dev, dtype=torch.bfloat16, generator=gen) * 0.5).to(FP8))
    return t


def rsc(*shape):
    return torch.rand(*shape, device=dev, generator=gen) * 0.02 + 0.01


def timeit(name, fns, nbytes, iters=a.iters):
    """Capture the launches into one hipGraph (what the decode step does) and time its replay:
    per-launch time = replay / len(fns), includes the in-graph kernel boundary."""
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    reps = max(1, 24 // len(fns))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            for f in fns:
                f()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / (reps * len(fns)))
    ts.sort()
    med = ts[len(ts) // 2]
    print(f"{name:58s} {med*1e3:8.2f} us  {nbytes/med/1e6:8.1f} GB/s  (min {ts[0]*1e3:.2f})", flush=True)


L = a.layers
E, K, I, topk = 257, 7168, 256, 9
want = lambda k: (not a.only) or any(o in k for o in a.only.split(","))
lib = _lib.lib()
if want("moe"):
    w1 = [rfp8(E, 2 * I, K) for _ in range(L)]
    w1s = [rsc(E, 4, 56) for _ in range(L)]
    w2 = [rfp8(E, K, I) for _ in range(L)]
    w2s = [rsc(E, 56, 2) for _ in range(L)]
    for bs in a.bs:
        ids = torch.stack([torch.randperm(E - 1, device=dev, generator=gen)[:topk - 1] for _ in range(bs)])
        ids = torch.cat([ids, torch.full((bs, 1), E - 1, device=dev)], 1).contiguous()
        distinct = int(ids.unique().numel())
        x = torch.randn(bs, K, device=dev, dtype=torch.bfloat16, generator=gen)
        xq, xs = fused_moe.per_token_group_quant_fp8(x, 128)
        sid, eid, npost = fused_moe.moe_align_block_size(ids, 16, E)
        numel = bs * topk
        c1 = torch.empty(numel, 2 * I, dtype=torch.bfloat16, device=dev)
        mmb = min(eid.numel(), numel)
        fns = [(lambda l=l: lib.chitu_hip_moe_gemm1_fp8(ptr(xq), ptr(xs), ptr(w1[l]), ptr(w1s[l]), ptr(sid), ptr(eid), ptr(npost),
                                                         ptr(c1), i64(numel), i32(topk), i64(2 * I), i64(K), i64(mmb), stream_ptr())) for l in range(L)]
        timeit(f"moe_gemm1 bs={bs} distinct={distinct} WK={os.environ.get('CHITU_MOE_GEMM1_WK','auto')} (apply with chitu_amd._lib.debug_option)", fns, distinct * 2 * I * K)
        hq, hs = fused_moe.silu_and_mul_quant(c1, mode="group")
        c3 = torch.empty(numel, K, dtype=torch.bfloat16, device=dev)
        wts = torch.rand(bs, topk, device=dev, generator=gen).to(torch.bfloat16)
        fns = [(lambda l=l: lib.chitu_hip_moe_gemm2_fp8(ptr(hq), ptr(hs), ptr(w2[l]), ptr(w2s[l]), ptr(sid), ptr(eid), ptr(npost),
                                                         ptr(wts), i32(0), i32(1), ptr(c3), i64(numel), i64(K), i64(I), i64(mmb), stream_ptr())) for l in range(L)]
        timeit(f"moe_gemm2 bs={bs} distinct={distinct}", fns, distinct * K * I)
        hb = torch.randn(numel, I, device=dev, dtype=torch.bfloat16, generator=gen)
        fns = [(lambda l=l: lib.chitu_hip_moe_gemm1_silu_fp8(ptr(xq), ptr(xs), ptr(w1[l]), ptr(w1s[l]), ptr(sid), ptr(eid), ptr(npost),
                                                              ptr(hb), i64(numel), i32(topk), i64(I), i64(K), i64(mmb), stream_ptr())) for l in range(L)]
        timeit(f"moe_gemm1_silu bs={bs} distinct={distinct}", fns, distinct * 2 * I * K)
        fns = [(lambda l=l: lib.chitu_hip_moe_gemm2_quant_fp8(ptr(hb), ptr(w2[l]), ptr(w2s[l]), ptr(sid), ptr(eid), ptr(npost), ptr(wts),
                                                               i32(0), i32(1), ptr(c3), i64(numel), i64(K), i64(I), i64(mmb), f32(1e-10),
                                                               stream_ptr())) for l in range(L)]
        timeit(f"moe_gemm2_quant bs={bs} distinct={distinct}", fns, distinct * K * I)
        fns = [lambda: fused_moe.silu_and_mul_quant(c1, mode="group")]
        timeit(f"moe_silu_quant bs={bs}", fns, numel * 2 * I * 2)
        out = torch.empty(bs, K, dtype=torch.bfloat16, device=dev)
        fns = [lambda: lib.chitu_hip_moe_sum(ptr(c3), ptr(out), i64(bs), i32(topk), i64(K), stream_ptr())]
        timeit(f"moe_sum bs={bs}", fns, numel * K * 2)
        fns = [lambda: fused_moe.moe_align_block_size(ids, 16, E)]
        timeit(f"moe_align bs={bs}", fns, numel * 8)
    del w1, w2
if want("dense"):
    shapes = {"wqkv_a": (2112, 7168), "wq_b": (3072, 1536), "wo": (7168, 2048), "dense_w1w3": (4608, 7168), "dense_w2": (7168, 2304)}
    for nm, (N, Kd) in shapes.items():
        n_buf = max(2, min(24, (600 << 20) // (N * Kd)))
        ws_ = [rfp8(N, Kd) for _ in range(n_buf)]
        ss_ = [rsc((N + 127) // 128, Kd // 128) for _ in range(n_buf)]
        for bs in a.bs:
            x = torch.randn(bs, Kd, device=dev, dtype=torch.bfloat16, generator=gen)
            xq, xs = ops.act_quant_deepseek_v3(x)
            fns = [(lambda l=l: ops.fp8_gemm_deepseek_v3(xq, xs, ws_[l], ss_[l], out_dtype=torch.bfloat16)) for l in range(n_buf)]
            timeit(f"fp8_gemm {nm} [{N}x{Kd}] bs={bs}", fns, N * Kd, iters=max(2, a.iters * 8 // n_buf))
        del ws_
if want("small"):
    for bs in a.bs:
        x = torch.randn(bs, 7168, device=dev, dtype=torch.bfloat16, generator=gen)
        add = torch.randn(bs, 7168, device=dev, dtype=torch.bfloat16, generator=gen)
        w = torch.ones(7168, device=dev, dtype=torch.bfloat16)
        timeit(f"rmsnorm+add+quant bs={bs}", [lambda: ops.rms_norm(x, w, 1e-6, out_bf16=True, quant="group", add=add)], bs * 7168 * 6)
        gw = (torch.randn(256, 7168, device=dev, generator=gen) * 0.01).to(torch.bfloat16)
        gb = (torch.randn(256, device=dev, generator=gen) * 0.01).to(torch.bfloat16)
        timeit(f"gate (scores+route) bs={bs}", [lambda: ops.gate_deepseek_v3(x, gw, gb, 8, 4, 8, "sigmoid", 2.5, extra_expert_id=256)], 256 * 7168 * 2)
        H, C = 16, 512
        wkv = rfp8(H * 256, C); sc = rsc(H * 2, 4)
        o = torch.randn(bs, H, C, device=dev, dtype=torch.bfloat16, generator=gen)
        w_uv = wkv.view(H, 256, C)[:, 128:]
        timeit(f"absorb_uv_quant bs={bs}", [lambda: ops.absorb_uv_quant_fp8(o, w_uv, sc, 4, 8, 1)], H * 128 * C)
        q = torch.randn(bs, H, 192, device=dev, dtype=torch.bfloat16, generator=gen)
        w_uk_t = wkv.view(torch.uint8).view(H, 256, C)[:, :128].transpose(1, 2).contiguous().view(FP8)
        timeit(f"absorb_uk bs={bs}", [lambda: ops.absorb_bmm_fp8(q[..., :128], w_uk_t, sc, 0, 8, 1, 0)], H * 128 * C)
if want("mla"):
    from chitu_amd.attn_backend import HipAttnBackend
    be = HipAttnBackend(16)
    for bs in a.bs:
        for ctx in (1024, 8192):
            pages = bs * (ctx // 64 + 1)
            caches = [torch.randn(pages, 64, 576, device=dev, dtype=torch.bfloat16, generator=gen) for _ in range(max(2, min(L, (1 << 30) // (pages * 64 * 576 * 2))))]
            table = torch.arange(pages, device=dev, dtype=torch.int32).view(bs, -1)
            sl = torch.full((bs,), ctx, device=dev, dtype=torch.int32)
            qn = torch.randn(bs, 16, 512, device=dev, dtype=torch.bfloat16, generator=gen)
            qp = torch.randn(bs, 16, 64, device=dev, dtype=torch.bfloat16, generator=gen)
            be.prepare_metadata_for_decode(sl, sl, table, 64)
            fns = [(lambda c=c: be.mla_decode(qn, qp, c, sl, table, 0.1352)) for c in caches]
            timeit(f"mla_decode(+merge) bs={bs} ctx={ctx} splits={be.num_splits}", fns, bs * ctx * 576 * 2)
