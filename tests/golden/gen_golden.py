"""Generate golden input/output fixtures by running the REFERENCE's own code on CPU.

Run in the build container only:   python tests/golden/gen_golden.py [name ...]
Writes tests/golden/<name>.npz.  The reference kernels are Triton; they run here
under TRITON_INTERPRET=1 (numpy), see ref_shims.py.  Known interpreter limits
(SURVEY.md 8c): bf16 tl.dot is broken (so MLA decode / bf16 MoE goldens are fed
bf16-representable values held in fp32), autotuned kernels must be called through
`kernel.fn[grid]` with one explicit config.
fp8 / bf16 arrays are stored as raw uint8 / uint16 bits.
"""

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402


def bits16(t):
    return t.contiguous().view(torch.int16).numpy().view(np.uint16)


def bits8(t):
    return t.contiguous().view(torch.uint8).numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in arrs.items()})


# ---------------------------------------------------------------- moe_align
def gen_moe_align():
    from chitu.fused_moe import moe_align_block_size_native

    cases = {}
    # the reference's own known-answer vector, fused_moe.py:478-487 (ids reach 4 => E=5)
    doc = torch.tensor([[2, 3, 4], [1, 2, 4], [1, 3, 4], [1, 2, 3]], dtype=torch.int64)
    cases["doc"] = (doc, 4, 5)
    g = torch.Generator().manual_seed(0)
    # the reference test's configuration, test/pytest/test_moe_align.py:11-14
    cases["reftest"] = (torch.randint(0, 256, (1000,), generator=g), 64, 256)
    # DeepSeek-R1 decode, bs=16, top-8 of 256, BLOCK_SIZE_M=64 (fused_moe.py:904-914)
    ids = torch.stack([torch.randperm(256, generator=g)[:8] for _ in range(16)])
    cases["r1_bs16"] = (ids, 64, 256)
    # block 16 (our grouped-GEMM tile), skewed routing
    cases["skew_b16"] = ((torch.randint(0, 1000, (300,), generator=g) % 7) * 3, 16, 32)
    out = {}
    for name, (ids, block, E) in cases.items():
        s, e, n = moe_align_block_size_native(ids, block, E)
        out[f"{name}_ids"] = ids.numpy()
        out[f"{name}_cfg"] = np.array([block, E], dtype=np.int64)
        out[f"{name}_sorted"] = s.numpy()
        out[f"{name}_experts"] = e.numpy()
        out[f"{name}_npost"] = n.numpy()
    save("moe_align", **out)


# ---------------------------------------------------------------- act quant + fp8 gemm
def gen_fp8_linear():
    import triton
    from chitu import ops
    from chitu.triton_kernels import fp8_gemm_deepseek_v3_kernel

    g = torch.Generator().manual_seed(1)
    M, N, K = 5, 384, 512
    x = (torch.randn(M, K, generator=g) * 0.7).to(torch.bfloat16)
    x[3, 128:256] *= 40.0  # one hot group
    xq, xs = ops.act_quant_deepseek_v3(x, 128)
    w = (torch.randn(N, K, generator=g) * 0.5).to(torch.float8_e4m3fn)
    ws = torch.rand(N // 128, K // 128, generator=g) * 0.02 + 0.01

    torch.set_default_dtype(torch.bfloat16)
    try:
        c = x.new_empty(M, N, dtype=torch.bfloat16)
        grid = (triton.cdiv(M, 16), triton.cdiv(N, 32))
        fp8_gemm_deepseek_v3_kernel.fn[grid](
            xq, w, c, xs, ws, M, N, K, group_n=128, group_k=128,
            BLOCK_SIZE_M=16, BLOCK_SIZE_N=32, BLOCK_SIZE_K=128,
        )
        wd = ops.weight_dequant_deepseek_v3(w, ws, 128)
    finally:
        torch.set_default_dtype(torch.float32)
    save(
        "fp8_linear",
        x=bits16(x), xq=bits8(xq), xs=xs.numpy(), w=bits8(w), ws=ws.numpy(),
        c=bits16(c), w_dequant=bits16(wd),
    )


# ---------------------------------------------------------------- per-token-group quant (MoE flavour)
def gen_group_quant():
    from chitu.fused_moe import per_token_group_quant_fp8

    g = torch.Generator().manual_seed(2)
    x = (torch.randn(6, 384, generator=g) * 1.3).to(torch.bfloat16)
    x[2, :128] = 0  # all-zero group: eps path (fused_moe.py:701)
    q, s = per_token_group_quant_fp8(x, 128)
    save("group_quant", x=bits16(x), q=bits8(q), s=s.numpy())


# ---------------------------------------------------------------- paged append + rope
def pattern_cache(pages, page, dim):
    """Deterministic cache fill, recomputed (not stored) by the tests."""
    p = torch.arange(pages).view(-1, 1, 1)
    s = torch.arange(page).view(1, -1, 1)
    d = torch.arange(dim).view(1, 1, -1)
    return (((p * 64 + s) % 251).float() * 0.25 + (d % 7).float() - 3.0).to(torch.bfloat16)


def gen_append_rope():
    from chitu import ops

    g = torch.Generator().manual_seed(3)
    pages, page, dim, bs = 12, 64, 576, 3
    cache = pattern_cache(pages, page, dim)
    table = torch.tensor([[3, 7, 0], [5, 1, 9], [2, 11, 4]], dtype=torch.int32)
    lens = torch.tensor([0, 64, 130], dtype=torch.int32)
    kv = torch.randn(bs, 1, 1, dim, generator=g).to(torch.bfloat16)
    before = cache.clone()
    ops.append_to_paged_kv_cache(cache, table, kv, lens)

    q = torch.randn(bs, 4, 64, generator=g).to(torch.bfloat16)
    k = torch.randn(bs, 64, generator=g).to(torch.bfloat16)
    cos = torch.randn(bs, 32, generator=g)
    sin = torch.randn(bs, 32, generator=g)
    oq, ok = ops.apply_rotary_pos_emb_torch(q, k, cos, sin, rotary_type="llama")
    tq, tk = ops.apply_rotary_pos_emb_triton(q, k, cos, sin, rotary_type="llama")
    changed = (cache != before).any(dim=-1).nonzero()  # [n, 2] (page, slot)
    save(
        "append_rope",
        cache_shape=np.array([pages, page, dim]), changed=changed.numpy(),
        changed_rows=bits16(cache[changed[:, 0], changed[:, 1]]), table=table.numpy(),
        lens=lens.numpy(), kv=bits16(kv),
        q=bits16(q), k=bits16(k), cos=cos.numpy(), sin=sin.numpy(),
        oq_torch=bits16(oq), ok_torch=bits16(ok), oq_triton=bits16(tq), ok_triton=bits16(tk),
    )


# ---------------------------------------------------------------- fused MoE (fp8 block w8a8)
def gen_fused_moe():
    from chitu.fused_moe import fused_experts_impl

    g = torch.Generator().manual_seed(4)
    M, E, topk, K, I = 6, 8, 2, 256, 128
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * I, K, generator=g) * 0.5).to(torch.float8_e4m3fn)
    w2 = (torch.randn(E, K, I, generator=g) * 0.5).to(torch.float8_e4m3fn)
    w1s = torch.rand(E, 2 * I // 128, K // 128, generator=g) * 0.02 + 0.01
    w2s = torch.rand(E, K // 128, I // 128, generator=g) * 0.02 + 0.01
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)])
    wts = torch.rand(M, topk, generator=g).to(torch.bfloat16)
    out = fused_experts_impl(
        x.clone(), w1, w2, wts, ids, inplace=False, use_fp8_w8a8=True,
        w1_scale=w1s, w2_scale=w2s, block_shape=[128, 128],
    )
    save(
        "fused_moe_fp8", x=bits16(x), w1=bits8(w1), w2=bits8(w2), w1s=w1s.numpy(), w2s=w2s.numpy(),
        ids=ids.numpy(), wts=bits16(wts), out=bits16(out),
    )


def gen_fused_moe_bf16():
    """The unquantised branch of the reference's fused MoE (use_fp8_w8a8=False), run in fp16 (rounding points live:
    GEMM outputs, SiluAndMul and the top-k sum round to fp16; the interpreter's float -> fp16 cast is numpy's RNE) and
    in fp32 (pure algorithm).  bf16 tl.dot is broken in the Triton interpreter, so bf16 itself cannot be
    generated; the oracle is dtype-generic and pinned on these two."""
    from chitu.fused_moe import fused_experts_impl

    g = torch.Generator().manual_seed(14)
    M, E, topk, K, I = 7, 8, 3, 256, 128
    x32 = (torch.randn(M, K, generator=g) * 0.5).to(torch.float16).float()
    w1_32 = (torch.randn(E, 2 * I, K, generator=g) * 0.1).to(torch.float16).float()
    w2_32 = (torch.randn(E, K, I, generator=g) * 0.1).to(torch.float16).float()
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)])
    wts32 = torch.rand(M, topk, generator=g).to(torch.float16).float()
    out16 = fused_experts_impl(x32.half(), w1_32.half(), w2_32.half(), wts32.half(), ids, inplace=False, use_fp8_w8a8=False)
    out32 = fused_experts_impl(x32.clone(), w1_32, w2_32, wts32, ids, inplace=False, use_fp8_w8a8=False)
    save("fused_moe_bf16", x=x32.numpy(), w1=w1_32.numpy(), w2=w2_32.numpy(), ids=ids.numpy(), wts=wts32.numpy(),
         out16=out16.float().numpy(), out32=out32.numpy())


# ---------------------------------------------------------------- MLA paged decode (fp32-held bf16 values)
def gen_mla_decode():
    from chitu.triton_decode_attention import _mla_attn_kernel, _mla_softmax_reducev

    g = torch.Generator().manual_seed(5)
    bs, H, C, R, page, pages = 3, 16, 512, 64, 64, 10
    lens = torch.tensor([1, 77, 200], dtype=torch.int32)  # incl. this token
    table = torch.tensor([[4, 0, 0, 0], [2, 9, 0, 0], [7, 1, 5, 3]], dtype=torch.int32)
    cache = (torch.randn(pages, page, C + R, generator=g)).to(torch.bfloat16).float()
    q_nope = (torch.randn(bs, H, C, generator=g) * 0.3).to(torch.bfloat16).float()
    q_pe = (torch.randn(bs, H, R, generator=g) * 0.3).to(torch.bfloat16).float()
    scale = 0.1352
    splits = 4
    logits = torch.zeros(bs, H, splits, C + 1)
    o = torch.zeros(bs, H, C)
    kv_c, k_pe = cache[..., :C], cache[..., C:]
    grid = (bs, 1, splits)
    _mla_attn_kernel.fn[grid](
        q_nope, q_pe, kv_c, k_pe, table, lens, logits, scale,
        q_nope.stride(0), q_nope.stride(1), q_pe.stride(0), q_pe.stride(1),
        kv_c.stride(-2), k_pe.stride(-2), table.stride(0),
        logits.stride(0), logits.stride(1), logits.stride(2),
        BLOCK_H=16, BLOCK_N=64, NUM_KV_SPLITS=splits, PAGE_SIZE=page,
        HEAD_DIM_CKV=C, HEAD_DIM_KPE=R,
    )
    _mla_softmax_reducev(logits, o, lens, splits)
    save(
        "mla_decode", cache=bits16(cache.to(torch.bfloat16)), q_nope=bits16(q_nope.to(torch.bfloat16)),
        q_pe=bits16(q_pe.to(torch.bfloat16)), table=table.numpy(), lens=lens.numpy(),
        scale=np.array([scale], dtype=np.float32), out=o.numpy(),
    )


# ---------------------------------------------------------------- GQA decode + MLA prefill (RefAttnBackend, pure torch)
def lattice(*shape, mod=97, scale=32.0, salt=0):
    """Deterministic bf16-exact values in [-1.5, 1.5] (multiples of 1/32), recomputed by the tests:
    v[i0, i1, ...] = ((sum_k i_k * prime_k + salt) % mod - mod//2) / scale."""
    primes = [131, 7, 53, 3, 17]
    acc = torch.zeros(shape, dtype=torch.int64) + salt
    for ax, n in enumerate(shape):
        view = [1] * len(shape)
        view[ax] = n
        acc = acc + torch.arange(n).view(view) * primes[ax]
    return ((acc % mod - mod // 2).float() / scale).to(torch.bfloat16)


def gen_gqa_decode():
    """RefAttnBackend.attn_with_kvcache (attn_backend.py:457-516) on CONTIGUOUS caches [B, S, Hkv, D]: the
    reference has no paged pure-torch path (block_table is rejected, :473), so the fixture pins the
    arithmetic (in-place append at cache_seqlens, GQA head mapping, softmax) on the same logical content
    that the tests lay out in pages."""
    from chitu.attn_backend import RefAttnBackend

    be = RefAttnBackend()
    B, S, Hq, Hkv, D = 4, 320, 8, 2, 128
    lens = torch.tensor([0, 255, 256, 300], dtype=torch.int64)  # empty cache, page edge (256), second page
    k_cache = lattice(B, S, Hkv, D, salt=1)
    v_cache = lattice(B, S, Hkv, D, salt=5)
    q = lattice(B, 1, Hq, D, mod=89, scale=64.0, salt=11)
    k_new = lattice(B, 1, Hkv, D, mod=83, salt=3)
    v_new = lattice(B, 1, Hkv, D, mod=79, salt=9)
    kc, vc = k_cache.clone(), v_cache.clone()
    out = be.attn_with_kvcache(q, kc, vc, k_new, v_new, cache_seqlens=lens, softmax_scale=D ** -0.5)
    for b in range(B):  # the reference appended in place
        assert torch.equal(kc[b, lens[b]], k_new[b, 0]) and torch.equal(vc[b, lens[b]], v_new[b, 0])
    save("gqa_decode", dims=np.array([B, S, Hq, Hkv, D], dtype=np.int64), lens=lens.numpy(), out=bits16(out))


def gen_mla_prefill():
    """RefAttnBackend.attn_varlen_func (attn_backend.py:394-455) called as AttentionDeepSeekV3.prefill_forward
    does in absorb mode (model_deepseek_v3.py:589-599): MQA, q [T, H, 576], k [T, 1, 576], v = k[..., :512]."""
    from chitu.attn_backend import RefAttnBackend

    be = RefAttnBackend()
    H, C, R = 16, 512, 64
    seqs = [1, 64, 65, 130, 7]
    T = sum(seqs)
    cu = torch.tensor([0] + list(np.cumsum(seqs)), dtype=torch.int32)
    kv = lattice(T, 1, C + R, salt=2)
    q = lattice(T, H, C + R, mod=89, scale=128.0, salt=13)
    out = be.attn_varlen_func(q, kv, kv[..., :C].contiguous(), cu, cu, max(seqs), max(seqs), causal=True, softmax_scale=0.1352)
    # keep the fixture small: first/last/page-edge tokens of every sequence + a stride through the rest
    keep = set()
    for s0, n in zip(cu[:-1].tolist(), seqs):
        keep.update(s0 + i for i in (0, 1, 62, 63, 64, 65, 127, 128, n - 2, n - 1) if 0 <= i < n)
    keep.update(range(0, T, 29))
    rows = np.array(sorted(keep), dtype=np.int64)
    save("mla_prefill", seqs=np.array(seqs, dtype=np.int64), dims=np.array([H, C, R], dtype=np.int64),
         scale=np.array([0.1352], dtype=np.float32), rows=rows, out=bits16(out[rows]))


def gen_gqa_prefill():
    """RefAttnBackend.attn_varlen_func as Attention.prefill_forward calls it (models/model.py:104-132):
    causal GQA, q [T, 8, 128], k / v [T, 2, 128]."""
    from chitu.attn_backend import RefAttnBackend

    be = RefAttnBackend()
    seqs = [1, 255, 257, 9]
    T = sum(seqs)
    cu = torch.tensor([0] + list(np.cumsum(seqs)), dtype=torch.int32)
    q = lattice(T, 8, 128, mod=89, scale=64.0, salt=4)
    k = lattice(T, 2, 128, salt=6)
    v = lattice(T, 2, 128, mod=83, salt=8)
    out = be.attn_varlen_func(q, k, v, cu, cu, max(seqs), max(seqs), causal=True)
    keep = set()
    for s0, n in zip(cu[:-1].tolist(), seqs):
        keep.update(s0 + i for i in (0, 1, 127, 254, 255, 256, n - 1) if 0 <= i < n)
    keep.update(range(0, T, 37))
    rows = np.array(sorted(keep), dtype=np.int64)
    save("gqa_prefill", seqs=np.array(seqs, dtype=np.int64), rows=rows, out=bits16(out[rows]))


# ---------------------------------------------------------------- W8A8 int8 quantisers (pure torch in the reference)
def gen_w8a8_quant():
    """chitu/quantize/w8a8.py:18-35 quant_act / quant_weight, imported with the closed GEMM packages
    (w8a8gemm / w8a8gemv, imported at module top, :4-5) stubbed -- they are not called by the quantisers."""
    import types

    for name in ("w8a8gemm", "w8a8gemv"):
        sys.modules.setdefault(name, types.ModuleType(name))
    from chitu.quantize.w8a8 import quant_act, quant_weight

    g = torch.Generator().manual_seed(21)
    x = (torch.randn(6, 512, generator=g) * 3).to(torch.float16)
    x[2] = 0  # all-zero row: scale clamps to 1e-5 / 127
    x[3, 7] = 60000.0  # large outlier
    w = (torch.randn(40, 512, generator=g) * 0.2).to(torch.float16)
    qx, sx = quant_act(x.clone())
    qw, sw = quant_weight(w.clone())
    save("w8a8_quant", x=x.view(torch.int16).numpy().view(np.uint16), w=w.view(torch.int16).numpy().view(np.uint16),
         qx=qx.numpy(), sx=sx.numpy(), qw=qw.numpy(), sw=sw.numpy())


GENS = {
    "moe_align": gen_moe_align,
    "fp8_linear": gen_fp8_linear,
    "group_quant": gen_group_quant,
    "append_rope": gen_append_rope,
    "fused_moe": gen_fused_moe,
    "fused_moe_bf16": gen_fused_moe_bf16,
    "mla_decode": gen_mla_decode,
    "gqa_decode": gen_gqa_decode,
    "mla_prefill": gen_mla_prefill,
    "w8a8_quant": gen_w8a8_quant,
    "gqa_prefill": gen_gqa_prefill,
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(GENS)
    for n in names:
        print("==", n)
        GENS[n]()
