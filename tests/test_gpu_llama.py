"""Llama-family decode step (BASELINE config 2 as a parity-test case): HIP layer vs the oracle
composition, graph replay == eager, KV advancing over steps, silu_and_mul op."""

import pytest
import torch

from oracle import llama as ollama
from tests.util import assert_close, max_rel_to_peak

pytestmark = pytest.mark.gpu


def tiny_args(n_kv_heads=2):
    from chitu_amd.llama import LlamaArgs

    return LlamaArgs(dim=1024, n_layers=3, n_heads=8, n_kv_heads=n_kv_heads, vocab_size=2048, ffn_dim=2048)


def build(args, max_reqs=4, max_seq=1024, page=256):
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.llama import LlamaDecoder, init_synthetic_

    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=max_reqs, block_size=page, max_seq_len=max_seq, device="cuda",
                                n_local_kv_heads=args.n_kv_heads, head_dim=args.head_dim, dtype=torch.bfloat16)
    model = LlamaDecoder(args, cache, HipAttnBackend(local_n_heads=args.n_heads, max_seq_len=max_seq),
                         max_position_embeddings=max_seq, device="cuda")
    init_synthetic_(model, seed=0)
    return model, cache


def test_silu_and_mul_bit_exact():
    from chitu_amd import ops

    g = torch.Generator().manual_seed(0)
    x = (torch.randn(7, 2 * 1408, generator=g) * 3).to(torch.bfloat16)
    ref = torch.nn.functional.silu(x[:, :1408]) * x[:, 1408:]
    out = ops.silu_and_mul(x.cuda()).cpu()
    d = (out.view(torch.int16).int() - ref.view(torch.int16).int()).abs()
    assert d.max() <= 1 and (d > 0).float().mean() < 0.01  # expf vs torch's exp: last-bit differences only


@pytest.mark.parametrize("n_kv_heads", [2, 8], ids=["gqa4", "mha"])
def test_layerwise_parity_and_graph_replay(n_kv_heads):
    args = tiny_args(n_kv_heads)
    model, cache = build(args)
    params = {k: v.detach().cpu() for k, v in model.named_parameters()}
    bs, reqs = 3, ["r0", "r1", "r2"]
    gen = torch.Generator().manual_seed(7)
    for r, n in zip(reqs, (0, 255, 300)):
        cache.register_sequence(r, n)
        for blk in cache.block_table[r]:
            cache.paged_k_cache[:, blk] = (torch.randn(args.n_layers, 256, n_kv_heads, 128, generator=gen) * 0.5).to(torch.bfloat16).cuda()
            cache.paged_v_cache[:, blk] = (torch.randn(args.n_layers, 256, n_kv_heads, 128, generator=gen) * 0.5).to(torch.bfloat16).cuda()
    shadow_k, shadow_v = cache.paged_k_cache.cpu().clone(), cache.paged_v_cache.cpu().clone()
    cache.prepare_cache_decode(reqs)
    cache.prepare_block_table_for_decode(reqs)
    lens = cache.get_gpu_seq_lens_excl_this_decode()[:bs].cpu()
    table = cache.get_gpu_block_table()[:bs].cpu()
    cos, sin = model.cos_table.cpu()[lens.long()], model.sin_table.cpu()[lens.long()]
    x = torch.randn(bs, args.dim, generator=gen).to(torch.bfloat16)
    worst = 0.0
    for i, layer in enumerate(model.layers):
        with torch.inference_mode():
            xm, pend = layer(x.cuda(), None, cos.cuda(), sin.cuda())
        y = (xm + pend).cpu()
        y_ref, k_new, v_new = ollama.block(params, f"layers.{i}.", x, cos, sin, shadow_k[i], shadow_v[i], table, lens,
                                           args.n_heads, n_kv_heads, 128, args.norm_eps)
        assert_close(cache.paged_k_cache[i].cpu(), k_new, 1e-2)
        assert_close(cache.paged_v_cache[i].cpu(), v_new, 1e-2)
        err = max_rel_to_peak(y, y_ref)
        worst = max(worst, err)
        assert err < 2e-2, (i, err)
        x = y_ref
    print("worst layer rel err", worst)

    tokens = torch.tensor([5, 17, 900], dtype=torch.int64, device="cuda")
    outs = []
    for step in range(3):
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        snap_k, snap_v = cache.paged_k_cache.clone(), cache.paged_v_cache.clone()
        eager = model.decode(tokens, use_graph=False).clone()
        kv_eager = (cache.paged_k_cache.clone(), cache.paged_v_cache.clone())
        cache.paged_k_cache.copy_(snap_k)
        cache.paged_v_cache.copy_(snap_v)
        graph = model.decode(tokens, use_graph=True).clone()
        assert torch.equal(eager, graph)
        assert torch.equal(kv_eager[0], cache.paged_k_cache) and torch.equal(kv_eager[1], cache.paged_v_cache)
        if step == 2:  # the N > 1 replay form (graph pieces cut at the collectives) on one rank
            cache.paged_k_cache.copy_(snap_k)
            cache.paged_v_cache.copy_(snap_v)
            assert torch.equal(eager, model.decode(tokens, use_graph="piecewise"))
            assert torch.equal(kv_eager[0], cache.paged_k_cache) and torch.equal(kv_eager[1], cache.paged_v_cache)
        assert eager.dtype == torch.float32 and tuple(eager.shape) == (bs, args.vocab_size) and torch.isfinite(eager).all()
        outs.append(eager)
        tokens = eager.argmax(dim=-1)
        cache.finalize_cache_single_decode(reqs)
    assert not torch.equal(outs[0], outs[1]) and len(model.graphs) == 2


@pytest.mark.parametrize("rotary", ["llama", "hf-llama"])
def test_qkv_post_equals_rope_plus_appends(rotary):
    from chitu_amd import ops

    g = torch.Generator().manual_seed(2)
    bs, hq, hkv, hd, pages = 5, 8, 2, 128, 12
    qkv = torch.randn(bs, hq + 2 * hkv, hd, generator=g).to(torch.bfloat16).cuda()
    cos, sin = torch.randn(bs, hd // 2, generator=g).cuda(), torch.randn(bs, hd // 2, generator=g).cuda()
    kc = torch.randn(pages, 256, hkv, hd, generator=g).to(torch.bfloat16).cuda()
    vc = torch.randn(pages, 256, hkv, hd, generator=g).to(torch.bfloat16).cuda()
    table = torch.stack([torch.randperm(pages, generator=g)[:2] for _ in range(bs)]).to(torch.int32).cuda()
    lens = torch.tensor([0, 255, 256, 300, 511], dtype=torch.int32).cuda()
    q_ref, k_ref = ops.apply_rotary_pos_emb(qkv[:, :hq], qkv[:, hq : hq + hkv], cos, sin, rotary_type=rotary)
    kc_ref, vc_ref = kc.clone(), vc.clone()
    ops.append_to_paged_kv_cache(kc_ref, table, k_ref.view(bs, 1, hkv, hd).contiguous(), lens)
    ops.append_to_paged_kv_cache(vc_ref, table, qkv[:, hq + hkv :].reshape(bs, 1, hkv, hd).contiguous(), lens)
    work = qkv.clone()
    q = ops.gqa_qkv_post(work, hq, hkv, cos, sin, kc, vc, table, lens, rotary_type=rotary)
    assert torch.equal(q, q_ref) and torch.equal(kc, kc_ref) and torch.equal(vc, vc_ref)
    assert torch.equal(work[:, hq:], qkv[:, hq:])  # k / v parts of the row untouched


@pytest.mark.parametrize("M,inter,K", [(1, 14336, 4096), (16, 14336, 4096), (20, 2048, 1024), (3, 1000, 512), (33, 256, 256)])
def test_gate_up_gemm_with_silu_epilogue_equals_gemm_then_silu(M, inter, K):
    from chitu_amd import ops

    g = torch.Generator().manual_seed(M + inter)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w13 = (torch.randn(2 * inter, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    ref = ops.silu_and_mul(ops.bf16_linear(x, w13))
    out = ops.bf16_linear_silu(x, w13)
    # same arithmetic; the K range may be split over a different number of waves (fp32 summation order),
    # so equality is up to the last bf16 bit of a few elements
    assert_close(out, ref, 4e-3)
    assert (out != ref).float().mean() < 0.05
    h13 = torch.nn.functional.linear(x.cpu().float(), w13.cpu().float()).to(torch.bfloat16)
    cpu = torch.nn.functional.silu(h13[:, :inter]) * h13[:, inter:]
    assert_close(out, cpu, 1e-2)


@pytest.mark.parametrize("M", [1, 2, 3, 4])
@pytest.mark.parametrize("N,inter,K", [(6144, 14336, 4096), (12288, 11008, 4096), (256, 2048, 1024), (100, 200, 512), (1024, 1024, 8192)])
def test_add_norm_as_gemm_prologue_is_bit_identical_to_the_separate_launches(M, N, inter, K):
    """bf16_linear_add_norm / bf16_linear_silu_add_norm (residual add + RMSNorm redone by every workgroup of the GEMM
    that consumes it) == rms_norm(x, add=...) followed by bf16_linear / bf16_linear_silu, bit for bit: new residual
    stream, projection, SwiGLU output.  Shapes: Llama-3-8B and Llama-2-7B layers, small ones, the 8192-wide limit."""
    from chitu_amd import ops

    if not ops.bf16_add_norm_fits(M, min(N, inter), K) or not ops.bf16_add_norm_fits(M, max(N, inter), K):
        with pytest.raises(AssertionError):
            ops.bf16_linear_add_norm(torch.zeros(M, K, dtype=torch.bfloat16, device="cuda"), torch.zeros(M, K, dtype=torch.bfloat16, device="cuda"),
                                     torch.ones(K, dtype=torch.bfloat16, device="cuda"), 1e-5,
                                     torch.zeros(N if not ops.bf16_add_norm_fits(M, N, K) else inter, K, dtype=torch.bfloat16, device="cuda"))
        return
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M + 2, K, generator=g) * 2).to(torch.bfloat16).cuda()[1 : M + 1]  # a view: row stride K, offset base
    add = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    nw = (1 + 0.2 * torch.randn(K, generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    w13 = (torch.randn(2 * inter, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    x_ref, y = ops.rms_norm(x, nw, 1e-5, add=add)
    o_ref, h_ref = ops.bf16_linear(y, w), ops.bf16_linear_silu(y, w13)
    x1, o = ops.bf16_linear_add_norm(x, add, nw, 1e-5, w)
    x2, h = ops.bf16_linear_silu_add_norm(x, add, nw, 1e-5, w13)
    assert torch.equal(x1, x_ref) and torch.equal(x2, x_ref)
    assert torch.equal(o, o_ref), (o.float() - o_ref.float()).abs().max().item()
    assert torch.equal(h, h_ref), (h.float() - h_ref.float()).abs().max().item()
    o32 = ops.bf16_linear_add_norm(x, add, nw, 1e-5, w, out_dtype=torch.float32)[1]
    assert torch.equal(o32, ops.bf16_linear(y, w, out_dtype=torch.float32))


@pytest.mark.parametrize("M", [1, 2, 4])
@pytest.mark.parametrize("hq,hkv,d,K", [(32, 8, 128, 4096), (32, 32, 128, 4096), (8, 2, 128, 1024), (4, 4, 64, 512)])
def test_qkv_projection_with_norm_prologue_and_rope_append_epilogue(M, hq, hkv, d, K):
    """bf16_linear_add_norm_qkv_post == rms_norm(add=) + bf16_linear + gqa_qkv_post (rotary "llama"), bit for bit:
    residual stream, rotated q heads, both caches (ragged lengths across page boundaries)."""
    from chitu_amd import ops

    g = torch.Generator().manual_seed(M + hq + K)
    x = (torch.randn(M, K, generator=g) * 2).to(torch.bfloat16).cuda()
    add = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    nw = (1 + 0.2 * torch.randn(K, generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn((hq + 2 * hkv) * d, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    cos, sin = torch.randn(M, d // 2, generator=g).cuda(), torch.randn(M, d // 2, generator=g).cuda()
    pages, page = 2 * M + 1, 16
    kc = torch.randn(pages, page, hkv, d, generator=g).to(torch.bfloat16).cuda()
    vc = torch.randn(pages, page, hkv, d, generator=g).to(torch.bfloat16).cuda()
    table = torch.randperm(pages, generator=g)[: 2 * M].view(M, 2).to(torch.int32).cuda()
    lens = torch.tensor([(11 * i + (15 if i % 2 else 16)) % 32 for i in range(M)], dtype=torch.int32).cuda()
    x_ref, y = ops.rms_norm(x, nw, 1e-5, add=add)
    qkv_ref = ops.bf16_linear(y, w).view(M, hq + 2 * hkv, d)
    k1, v1 = kc.clone(), vc.clone()
    q_ref = ops.gqa_qkv_post(qkv_ref, hq, hkv, cos, sin, k1, v1, table, lens, rotary_type="llama")
    k2, v2 = kc.clone(), vc.clone()
    x2, qkv = ops.bf16_linear_add_norm_qkv_post(x, add, nw, 1e-5, w, hq, hkv, cos, sin, k2, v2, table, lens)
    assert torch.equal(x2, x_ref) and torch.equal(qkv[:, :hq], q_ref)
    assert torch.equal(k1, k2) and torch.equal(v1, v2) and not torch.equal(k2, kc) and not torch.equal(v2, vc)


@pytest.mark.parametrize("bs", [1, 2])
def test_decode_with_norms_in_the_gemm_prologues_equals_decode_without(bs, monkeypatch):
    """A whole decode step (eager and graph replay) with the add + norm steps fused into the GEMMs behind them against
    the same step with every norm as its own launch: identical logits and KV pages."""
    from chitu_amd import llama

    args = tiny_args(2)
    model, cache = build(args)
    reqs = [f"q{i}" for i in range(bs)]
    gen = torch.Generator().manual_seed(11)
    for r, n in zip(reqs, (300, 17)):
        cache.register_sequence(r, n)
    cache.paged_k_cache.normal_(0, 0.5)
    cache.paged_v_cache.normal_(0, 0.5)
    tokens = torch.randint(0, args.vocab_size, (bs,), generator=gen).cuda()
    cache.prepare_cache_decode(reqs)
    cache.prepare_block_table_for_decode(reqs)
    snap_k, snap_v = cache.paged_k_cache.clone(), cache.paged_v_cache.clone()
    res = {}
    for fuse in (0, 4):
        monkeypatch.setattr(llama, "FUSE_NORM_MAX_BS", fuse)
        model.graphs.clear()
        for mode in (False, True):
            cache.paged_k_cache.copy_(snap_k)
            cache.paged_v_cache.copy_(snap_v)
            res[(fuse, mode)] = (model.decode(tokens, use_graph=mode).clone(), cache.paged_k_cache.clone(), cache.paged_v_cache.clone())
    base = res[(0, False)]
    for key, val in res.items():
        assert all(torch.equal(a, b) for a, b in zip(val, base)), key


@pytest.mark.parametrize("M,N,K", [(128, 1024, 7168), (2048, 256, 7168), (1000, 6144, 4096), (257, 200, 512), (300, 4096, 14336)])
def test_bf16_gemm_tiled_prefill_form_vs_streaming_form(M, N, K):
    """M >= 128 takes the compute-shaped bf16 kernel (bf16_gemm_tiled.hip); against the weight-streaming kernel forced on
    the same inputs: fp32 outputs agree to summation order (<= 1e-4 of the peak), bf16 outputs to their last bit, and the
    result is torch's F.linear up to that bit; ragged edges included; run to run identical."""
    from chitu_amd import _lib, ops

    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    tiled = ops.bf16_linear(x, w, out_dtype=torch.float32)
    with _lib.debug_option("bf16_gemm_tiled", 0):
        streamed = ops.bf16_linear(x, w, out_dtype=torch.float32)
    assert tuple(tiled.shape) == (M, N) and torch.isfinite(tiled).all()
    assert_close(tiled, streamed, 1e-4)
    assert torch.equal(tiled, ops.bf16_linear(x, w, out_dtype=torch.float32))
    ref = torch.nn.functional.linear(x.float(), w.float())
    assert_close(tiled, ref, 1e-4)
    assert_close(ops.bf16_linear(x, w), ref, 8e-3)# one bf16 ulp of the peak binade


@pytest.mark.parametrize("M,N,K,S", [(2048, 256, 7168, 8), (300, 200, 1024, 4), (256, 256, 7168, 16), (512, 128, 192, 3)])
def test_bf16_gemm_tiled_split_k_planes_sum_to_the_gemm(M, N, K, S):
    """The tiled kernel with the K range cut over S workgroups per tile (the router's score GEMM of a long prompt: few tiles,
    long K): the fp32 planes [S, M, N] summed in plane order equal the one-pass fp32 output to summation order, every plane
    is fully written (NaN-poisoned buffer), ragged tiles and a K block count that S does not divide included."""
    from chitu_amd import _lib, ops
    from chitu_amd._lib import check, i32, i64, ptr, stream_ptr

    g = torch.Generator().manual_seed(M + N + K + S)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    planes = torch.full((S, M, N), float("nan"), dtype=torch.float32, device="cuda")
    check(_lib.lib().chitu_hip_bf16_gemm(ptr(x), ptr(w), ptr(None), i32(0), i64(M), i64(N), i64(K), i32(S), ptr(planes), stream_ptr()),
          "bf16_gemm split-K")
    assert torch.isfinite(planes).all()
    total = planes[0].clone()
    for i in range(1, S):
        total += planes[i]
    assert_close(total, ops.bf16_linear(x, w, out_dtype=torch.float32), 1e-4)
    assert_close(total, torch.nn.functional.linear(x.float(), w.float()), 1e-4)


@pytest.mark.parametrize("M,N,K,S", [(128, 1024, 7168, 1), (1000, 6144, 4096, 1), (257, 200, 512, 1), (2048, 256, 7168, 8), (300, 200, 1024, 4)])
def test_bf16_gemm_tiled_token_tile_heights_return_the_same_bits(M, N, K, S):
    """Round 6: 64-token tiles for small grids (launcher heuristic; option fp8_tiled_tm forces either height for both tiled
    GEMMs).  The same arithmetic per output element in the same order: bit-identical outputs and split-K planes."""
    from chitu_amd import _lib, ops
    from chitu_amd._lib import check, i32, i64, ptr, stream_ptr

    g = torch.Generator().manual_seed(M + 2 * N + K + S)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    outs = {}
    for tm in (64, 128):
        with _lib.debug_option("fp8_tiled_tm", tm):
            if S == 1:
                outs[tm] = ops.bf16_linear(x, w, out_dtype=torch.float32)
            else:
                planes = torch.full((S, M, N), float("nan"), dtype=torch.float32, device="cuda")
                check(_lib.lib().chitu_hip_bf16_gemm(ptr(x), ptr(w), ptr(None), i32(0), i64(M), i64(N), i64(K), i32(S), ptr(planes), stream_ptr()),
                      "bf16_gemm split-K")
                outs[tm] = planes
    assert torch.isfinite(outs[64]).all() and torch.equal(outs[64], outs[128])


def test_prefill_equals_token_by_token_decode_and_generate():
    args = tiny_args(2)
    model, cache = build(args)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, args.vocab_size, (n,), generator=g).tolist() for n in (1, 260, 9)]
    reqs = ["d0", "d1", "d2"]
    for r in reqs:
        cache.register_sequence(r, 0)
    last_logits = [None] * 3
    for step in range(max(len(p) for p in prompts)):
        live = [i for i, p in enumerate(prompts) if step < len(p)]
        ids = [reqs[i] for i in live]
        cache.prepare_cache_decode(ids)
        cache.prepare_block_table_for_decode(ids)
        toks = torch.tensor([prompts[i][step] for i in live], dtype=torch.int64, device="cuda")
        logits = model.decode(toks, use_graph=False)
        cache.finalize_cache_single_decode(ids)
        for kk, i in enumerate(live):
            if step == len(prompts[i]) - 1:
                last_logits[i] = logits[kk].clone()
    for r in reqs:
        cache.finalize_cache_all_decode(r)
    logits_p = model.prefill(prompts, ["p0", "p1", "p2"])
    for i in range(3):
        assert_close(logits_p[i], last_logits[i], 3e-2, what=i)
    for r in ("p0", "p1", "p2"):
        cache.finalize_cache_all_decode(r)
    free_before = len(cache.free_blocks)
    out1, out2 = model.generate(prompts, 4), model.generate(prompts, 4)
    assert tuple(out1.shape) == (3, 4) and torch.equal(out1, out2) and len(cache.free_blocks) == free_before


@pytest.mark.parametrize("M", [1, 2])
@pytest.mark.parametrize("N,K", [(6144, 4096), (256, 1024), (1024, 8192)])
def test_add_norm_gemm_prologue_with_two_residual_terms(M, N, K):
    """ops.bf16_linear_add_norm(add=[M, 2, K]) (round 6: a top-2 MoE's un-summed outputs handed to the next layer's qkv GEMM) ==
    moe_sum of the two terms (one bf16 rounding) + rms_norm(add=) + bf16_linear, bit for bit -- and == rms_norm(add=<3-D>), the
    other consumer of the same hand-off."""
    from chitu_amd import ops

    if not ops.bf16_add_norm_fits(M, N, K):
        pytest.skip("shape outside the fused launch")
    g = torch.Generator().manual_seed(7 * M + N + K)
    x = (torch.randn(M, K, generator=g) * 2).to(torch.bfloat16).cuda()
    terms = torch.randn(M, 2, K, generator=g).to(torch.bfloat16).cuda()
    nw = (1 + 0.2 * torch.randn(K, generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(torch.bfloat16).cuda()
    summed = (terms[:, 0].float() + terms[:, 1].float()).to(torch.bfloat16)
    x_ref, y = ops.rms_norm(x, nw, 1e-5, add=summed)
    x_ref3, y3 = ops.rms_norm(x, nw, 1e-5, add=terms)
    assert torch.equal(x_ref, x_ref3) and torch.equal(y, y3)
    x1, o = ops.bf16_linear_add_norm(x, terms, nw, 1e-5, w)
    assert torch.equal(x1, x_ref) and torch.equal(o, ops.bf16_linear(y, w))
