"""End-to-end decode-layer parity: chitu_amd.deepseek_v3 (HIP ops) vs oracle.deepseek (CPU restatement
of the reference's model_deepseek_v3.py decode path), plus RMSNorm / absorb op tests and hipGraph replay."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import deepseek as ods
from oracle import fp8 as ofp8
from tests.util import assert_close, bits16, bits8, max_rel_to_peak

pytestmark = pytest.mark.gpu


def tiny_args():
    from chitu_amd.deepseek_v3 import DeepSeekV3Args

    return DeepSeekV3Args(
        vocab_size=1024, dim=512, inter_dim=1024, moe_inter_dim=256, n_layers=3, n_dense_layers=1, n_heads=16,
        n_routed_experts=16, n_shared_experts=1, n_activated_experts=4, n_expert_groups=4, n_limited_groups=2,
        q_lora_rank=256, gate_bias=True,
    )


def v2lite_like_args():
    """Structure of BASELINE config 3 (DeepSeek-V2-Lite) at test size: softmax scores, one
    expert group, no gate bias, TWO shared experts, top-6, moe_inter = 5 K-blocks (generic GEMM2 and
    the three-launch expert path), q_lora_rank = 0 (q = wq(x): the branch the reference's loader prepares,
    backend.py:460, but its attention asserts away, model_deepseek_v3.py:477 -- SURVEY gap G1)."""
    from chitu_amd.deepseek_v3 import DeepSeekV3Args

    return DeepSeekV3Args(
        vocab_size=1024, dim=512, inter_dim=1024, moe_inter_dim=640, n_layers=3, n_dense_layers=1, n_heads=16,
        n_routed_experts=16, n_shared_experts=2, n_activated_experts=6, n_expert_groups=1, n_limited_groups=1,
        q_lora_rank=0, gate_bias=False, score_func="softmax", route_scale=1.0,
    )


def build(args, max_reqs=4, max_seq=512):
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Decoder, init_synthetic_

    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=max_reqs, block_size=64, max_seq_len=max_seq,
                                device="cuda", kv_shape_per_sample=(576,), dtype=torch.bfloat16)
    be = HipAttnBackend(local_n_heads=args.n_heads, max_seq_len=max_seq)
    model = DeepSeekV3Decoder(args, cache, be, max_position_embeddings=max_seq, device="cuda")
    init_synthetic_(model, seed=0)
    return model, cache


def cfg_of(args):
    from chitu_amd.deepseek_v3 import compute_softmax_scale

    return dict(H=args.n_heads, C=args.kv_lora_rank, R=args.qk_rope_head_dim, NOPE=args.qk_nope_head_dim,
                V=args.v_head_dim, QL=args.q_lora_rank, eps=args.norm_eps, scale=compute_softmax_scale(args),
                n_groups=args.n_expert_groups, topk_groups=args.n_limited_groups, topk=args.n_activated_experts,
                score_func=args.score_func, route_scale=args.route_scale, n_routed=args.n_routed_experts)


@pytest.mark.parametrize("rows,dim", [(1, 7168), (16, 7168), (5, 1536), (3, 512), (2, 8192)])
def test_rmsnorm_matches_torch_and_fused_quant(rows, dim):
    from chitu_amd import ops

    g = torch.Generator().manual_seed(dim + rows)
    x = (torch.randn(rows, dim, generator=g) * 3).to(torch.bfloat16)
    w = (torch.rand(dim, generator=g) + 0.5).to(torch.bfloat16)
    ref = F.rms_norm(x, (dim,), w, 1e-6)
    y = ops.rms_norm(x.cuda(), w.cuda(), 1e-6)
    d = np.abs(bits16(y).astype(np.int32) - bits16(ref).astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.01  # same formula; rsqrt / summation order only
    for mode, oracle in (("act", ofp8.act_quant_deepseek_v3), ("group", ofp8.per_token_group_quant_fp8)):
        y2, q, s = ops.rms_norm(x.cuda(), w.cuda(), 1e-6, quant=mode)
        assert torch.equal(y2, y)
        q_ref, s_ref = oracle(y.cpu())  # quantisation of the *rounded* norm output
        assert np.array_equal(bits8(q), bits8(q_ref)) and np.array_equal(s.cpu().numpy(), s_ref.numpy())
    # strided input rows (q_a_kv[:, :q_lora] views)
    big = (torch.randn(rows, dim + 576, generator=g)).to(torch.bfloat16)
    yv = ops.rms_norm(big.cuda()[:, :dim], w.cuda(), 1e-6)
    refv = F.rms_norm(big[:, :dim], (dim,), w, 1e-6)
    assert np.abs(bits16(yv).astype(np.int32) - bits16(refv).astype(np.int32)).max() <= 1


@pytest.mark.parametrize("bs", [1, 16, 19])
def test_absorb_projections(bs):
    from chitu_amd import ops

    H, C = 16, 512
    g = torch.Generator().manual_seed(bs)
    wkv_b = (torch.randn(H * 256, C, generator=g) * 0.5).to(torch.float8_e4m3fn)
    sc = torch.rand(H * 2, C // 128, generator=g) * 0.02 + 0.01
    wd = ofp8.weight_dequant_deepseek_v3(wkv_b, sc).view(H, 256, C)
    q = (torch.randn(bs, H, 192, generator=g)).to(torch.bfloat16)
    q_nope = q[..., :128]
    ref_uk = torch.einsum("shd,hdc->shc", q_nope.float(), wd[:, :128].float())
    w_uk_t = wkv_b.view(torch.uint8).view(H, 256, C)[:, :128].transpose(1, 2).contiguous().view(torch.float8_e4m3fn)
    out = ops.absorb_bmm_fp8(q.cuda()[..., :128], w_uk_t.cuda(), sc.cuda(), 0, 8, 1, 0)
    assert tuple(out.shape) == (bs, H, C) and max_rel_to_peak(out, ref_uk) < 5e-3
    o = (torch.randn(bs, H, C, generator=g)).to(torch.bfloat16)
    ref_uv = torch.einsum("bhc,hdc->bhd", o.float(), wd[:, 128:].float())
    w_uv = wkv_b.cuda().view(H, 256, C)[:, 128:]
    out2 = ops.absorb_bmm_fp8(o.cuda(), w_uv, sc.cuda(), 4, 8, 0, 1)
    assert tuple(out2.shape) == (bs, H, 128) and max_rel_to_peak(out2, ref_uv) < 5e-3


@pytest.mark.parametrize("make_args", [tiny_args, v2lite_like_args], ids=["v3_like", "v2lite_like"])
def test_layerwise_parity_and_graph_replay(make_args):
    args = make_args()
    model, cache = build(args)
    cfg = cfg_of(args)
    params = {k: v.detach().cpu() for k, v in model.named_parameters()}
    bs = 3
    reqs = ["r0", "r1", "r2"]
    gen = torch.Generator().manual_seed(7)
    for r, n in zip(reqs, (0, 63, 130)):
        cache.register_sequence(r, n)
        for blk in cache.block_table[r]:
            cache.paged_kv_cache[:, blk] = (torch.randn(args.n_layers, 64, 576, generator=gen) * 0.5).to(torch.bfloat16).cuda()
    shadow = cache.paged_kv_cache.cpu().clone()

    cache.prepare_cache_decode(reqs)
    cache.prepare_block_table_for_decode(reqs)
    model.prepare_decoding_attn()
    lens_excl = cache.get_gpu_seq_lens_excl_this_decode().cpu()
    table = cache.get_gpu_block_table().cpu()
    cos, sin = model.cos_table.cpu()[lens_excl.long()], model.sin_table.cpu()[lens_excl.long()]

    x = (torch.randn(bs, args.dim, generator=gen)).to(torch.bfloat16)
    worst = 0.0
    for i, layer in enumerate(model.layers):
        routing = {}
        if layer.is_moe:
            orig = layer.ffn.gate.forward

            def hooked(inp, _orig=orig, _store=routing, **kw):
                res = _orig(inp, **kw)  # (weights, ids[, moe_align triple of the fused route + sort launch])
                w, idx = res[0], res[1]
                k = args.n_activated_experts
                ns = args.n_shared_experts  # shared-expert slots: ids n_routed .., weight 1
                assert idx.shape[1] == k + ns and (w[:, k:] == 1).all()
                assert torch.equal(idx[:, k:].cpu(), (args.n_routed_experts + torch.arange(ns)).expand(idx.shape[0], -1))
                _store["w"], _store["i"] = w[:, :k].cpu(), idx[:, :k].cpu()
                return res

            layer.ffn.gate.forward = hooked
        with torch.inference_mode():
            xm, pend = layer(x.cuda(), None, cos.cuda(), sin.cuda())
            if pend.dim() == 3:  # un-summed top-k terms (the next norm folds the sum in): moe_sum's arithmetic
                pend = pend.float().sum(1).to(torch.bfloat16)
            y = (xm + pend).cpu()
        if layer.is_moe:
            layer.ffn.gate.forward = orig
        rt = (routing["w"], routing["i"]) if layer.is_moe else None
        y_ref, new_cache, _ = ods.block(params, i, x, cos, sin, shadow[i], table, lens_excl, cfg, layer.is_moe, rt)
        # the appended KV row: same bf16 values up to one rounding of the fp8 GEMM/norm chain
        assert_close(cache.paged_kv_cache[i].cpu(), new_cache, 1e-2)
        err = max_rel_to_peak(y, y_ref)
        worst = max(worst, err)
        assert err < 2e-2, (i, err)
        if layer.is_moe:
            # gate parity on the oracle's own normalised input: same experts except bf16 ties
            hn = ods.rms_norm(x + ods.attention_decode(params, f"layers.{i}.attn.", ods.rms_norm(x, params[f"layers.{i}.attn_norm.weight"], cfg["eps"]),
                                                       cos, sin, shadow[i], table, lens_excl, cfg)[0],
                              params[f"layers.{i}.ffn_norm.weight"], cfg["eps"])
            w_ref, i_ref = ods.gate(hn, params[f"layers.{i}.ffn.gate.weight"], params.get(f"layers.{i}.ffn.gate.bias"),
                                    cfg["n_groups"], cfg["topk_groups"], cfg["topk"], cfg["score_func"], cfg["route_scale"])
            same = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(i_ref, rt[1])) / i_ref.numel()
            assert same >= 0.75
        x = y_ref  # feed the oracle's activations forward so errors do not compound across layers
    print("worst layer rel err", worst)

    # ---- full step: graph replay == eager, three consecutive steps, KV advancing
    tokens = torch.tensor([5, 17, 900], dtype=torch.int64, device="cuda")
    outs = []
    for step in range(3):
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        snap = cache.paged_kv_cache.clone()
        eager = model.decode(tokens, use_graph=False).clone()
        kv_eager = cache.paged_kv_cache.clone()
        cache.paged_kv_cache.copy_(snap)
        graph = model.decode(tokens, use_graph=True).clone()
        assert torch.equal(eager, graph)
        assert torch.equal(kv_eager, cache.paged_kv_cache)
        assert eager.dtype == torch.float32 and tuple(eager.shape) == (bs, args.vocab_size)
        assert torch.isfinite(eager).all()
        outs.append(eager)
        tokens = eager.argmax(dim=-1)
        cache.finalize_cache_single_decode(reqs)
    assert not torch.equal(outs[0], outs[1])
    assert len(model.graphs) == 1

    # ---- piecewise replay (the N > 1 form: the step cut at every collective, collectives between the pieces;
    # with one rank every cut carries no collective) == eager, KV advancing
    for step in range(2):
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        snap = cache.paged_kv_cache.clone()
        eager = model.decode(tokens, use_graph=False).clone()
        kv_eager = cache.paged_kv_cache.clone()
        cache.paged_kv_cache.copy_(snap)
        pieces = model.decode(tokens, use_graph="piecewise").clone()
        assert torch.equal(eager, pieces) and torch.equal(kv_eager, cache.paged_kv_cache)
        tokens = eager.argmax(dim=-1)
        cache.finalize_cache_single_decode(reqs)
    assert len(model.graphs[(bs, "piecewise")].pieces) > args.n_layers


def test_rmsnorm_with_residual_add():
    from chitu_amd import ops

    g = torch.Generator().manual_seed(1)
    x = (torch.randn(5, 7168, generator=g) * 2).to(torch.bfloat16)
    a = (torch.randn(5, 7168, generator=g)).to(torch.bfloat16)
    w = (torch.rand(7168, generator=g) + 0.5).to(torch.bfloat16)
    xs = x + a  # bf16 add, as in the reference block
    ref = F.rms_norm(xs, (7168,), w, 1e-6)
    s, y, q, sc = ops.rms_norm(x.cuda(), w.cuda(), 1e-6, quant="group", add=a.cuda())
    assert torch.equal(s.cpu(), xs)
    assert np.abs(bits16(y).astype(np.int32) - bits16(ref).astype(np.int32)).max() <= 1
    q_ref, s_ref = ofp8.per_token_group_quant_fp8(y.cpu())
    assert np.array_equal(bits8(q), bits8(q_ref)) and np.array_equal(sc.cpu().numpy(), s_ref.numpy())


@pytest.mark.parametrize("M,N,K", [(1, 256, 7168), (16, 256, 7168), (20, 16160, 7168), (3, 1000, 512)])
def test_bf16_linear(M, N, K):
    from chitu_amd import ops

    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    ref = x.float() @ w.float().T
    y = ops.bf16_linear(x.cuda(), w.cuda())
    assert_close(y, ref, 5e-3)
    y32 = ops.bf16_linear(x.cuda(), w.cuda(), out_dtype=torch.float32)
    assert_close(y32, ref, 1e-4)


@pytest.mark.parametrize("score_func,bias,groups", [("sigmoid", True, (8, 4)), ("sigmoid", False, (8, 4)),
                                                     ("softmax", False, (1, 1)), ("softmax", False, (4, 2))])
@pytest.mark.parametrize("M", [1, 16])
def test_gate_route_on_identical_logits(score_func, bias, groups, M):
    """Routing kernel vs the oracle's verbatim GateDeepSeekV3 tail on the SAME bf16 logits.  bf16
    scores tie often and torch.topk's tie order is unspecified, so the check is: the HIP selection is a
    valid top-k of the oracle's masked scores (nothing unselected beats a selected expert), and where
    the expert sets agree the weights agree exactly."""
    from chitu_amd import _lib
    from chitu_amd._lib import f32, i32, i64, ptr, stream_ptr

    E, topk = 256, 8
    g = torch.Generator().manual_seed(M * 7 + len(score_func) + (3 if bias else 0))
    logits = (torch.randn(M, E, generator=g) * 1.5).to(torch.bfloat16)
    b = (torch.randn(E, generator=g) * 0.05).to(torch.bfloat16) if bias else None
    w_ref, i_ref, mid = ods.gate_from_logits(logits, b, groups[0], groups[1], topk, score_func, 2.5, return_masked=True)
    w_hip = torch.empty(M, topk, dtype=torch.bfloat16, device="cuda")
    i_hip = torch.empty(M, topk, dtype=torch.int64, device="cuda")
    ld, bd = logits.cuda(), (b.cuda() if bias else None)
    rc = _lib.lib().chitu_hip_gate_route(ptr(ld), i32(0), i64(M), i32(E), ptr(bd), i32(groups[0]), i32(groups[1]),
                                         i32(topk), i32(1 if score_func == "sigmoid" else 0), f32(2.5), ptr(w_hip),
                                         ptr(i_hip), i32(topk), i32(-1), f32(0.0), i32(0), stream_ptr())
    assert rc == 0
    w_hip, i_hip = w_hip.cpu(), i_hip.cpu()
    # expected result = the reference's intermediate tensors + the documented tie rule (lower index first)
    pre = mid["pre_mask"].float().reshape(M, E)
    orig = mid["original"]
    exact_torch = 0
    for r in range(M):
        sel_scores = pre[r].clone()
        if groups[0] > 1:
            gsc = mid["group_scores"][r].float().tolist()
            keep = sorted(range(groups[0]), key=lambda gi: (-gsc[gi], gi))[: groups[1]]
            mask = torch.zeros(groups[0])
            mask[keep] = 1
            sel_scores = (sel_scores.view(groups[0], -1) * mask[:, None]).flatten()
        vals = sel_scores.tolist()
        exp_ids = sorted(range(E), key=lambda e: (-vals[e], e))[:topk]
        assert i_hip[r].tolist() == exp_ids
        wsel = orig[r][exp_ids]
        if score_func == "sigmoid":
            wsel = wsel / wsel.sum(dim=-1, keepdim=True)
        wsel = (wsel * 2.5).type_as(logits)
        assert torch.equal(w_hip[r], wsel)
        exact_torch += set(exp_ids) == set(i_ref[r].tolist())
    assert exact_torch >= M // 4  # torch.topk agrees wherever its (unspecified) tie order does not matter


@pytest.mark.parametrize("M", [1, 16])
def test_gate_end_to_end(M):
    """Score GEMM (split-K partials) + routing vs the oracle gate on low-tie data."""
    from chitu_amd import ops

    E, K, topk = 256, 7168, 8
    g = torch.Generator().manual_seed(M)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(E, K, generator=g) * 0.01).to(torch.bfloat16)
    b = (torch.randn(E, generator=g) * 0.02).to(torch.bfloat16)
    w_ref, i_ref = ods.gate(x, w, b, 8, 4, topk, "sigmoid", 2.5)
    w_hip, i_hip = ops.gate_deepseek_v3(x.cuda(), w.cuda(), b.cuda(), 8, 4, topk, "sigmoid", 2.5)
    overlap = sum(len(set(a.tolist()) & set(c.tolist())) for a, c in zip(i_ref, i_hip.cpu())) / i_ref.numel()
    assert overlap >= 0.85, overlap
    w2, i2 = ops.gate_deepseek_v3(x.cuda(), w.cuda(), b.cuda(), 8, 4, topk, "sigmoid", 2.5, extra_expert_id=E)
    assert torch.equal(i2[:, :topk], i_hip) and (i2[:, topk] == E).all() and (w2[:, topk] == 1).all()


def test_kv_prep_equals_separate_ops():
    """Fused kv_norm + RoPE + append == rms_norm, apply_rotary_pos_emb, cat, append_to_paged_kv_cache."""
    from chitu_amd import ops

    g = torch.Generator().manual_seed(4)
    bs, H = 5, 16
    q_a_kv = torch.randn(bs, 1536 + 576, generator=g).to(torch.bfloat16).cuda()
    q = torch.randn(bs, H, 192, generator=g).to(torch.bfloat16).cuda()
    cos, sin = torch.randn(bs, 32, generator=g).cuda(), torch.randn(bs, 32, generator=g).cuda()
    wn = (torch.rand(512, generator=g) + 0.5).to(torch.bfloat16).cuda()
    cache = torch.randn(12, 64, 576, generator=g).to(torch.bfloat16).cuda()
    table = torch.stack([torch.randperm(12, generator=g)[:2] for _ in range(bs)]).to(torch.int32).cuda()
    lens = torch.tensor([0, 63, 64, 100, 127], dtype=torch.int32).cuda()
    # separate ops
    q_pe_ref, k_pe_ref = ops.apply_rotary_pos_emb(q[..., 128:], q_a_kv[:, 2048:], cos, sin, "llama")
    kvn = ops.rms_norm(q_a_kv[:, 1536:2048], wn, 1e-6)
    cache_ref = cache.clone()
    ops.append_to_paged_kv_cache(cache_ref, table, torch.cat([kvn, k_pe_ref], -1).view(bs, 1, 1, 576), lens)
    # fused
    q2 = q.clone()
    cache2 = cache.clone()
    ops.mla_kv_prep(q_a_kv[:, 1536:], q2[..., 128:], cos, sin, wn, 1e-6, cache2, table, lens)
    assert torch.equal(cache2, cache_ref)
    assert torch.equal(q2[..., 128:], q_pe_ref) and torch.equal(q2[..., :128], q[..., :128])


@pytest.mark.parametrize("bs", [1, 16, 21])
def test_absorb_uv_quant_equals_absorb_then_act_quant(bs):
    from chitu_amd import ops

    H, C = 16, 512
    g = torch.Generator().manual_seed(bs + 40)
    wkv_b = (torch.randn(H * 256, C, generator=g) * 0.5).to(torch.float8_e4m3fn).cuda()
    sc = (torch.rand(H * 2, C // 128, generator=g) * 0.02 + 0.01).cuda()
    o = torch.randn(bs, H, C, generator=g).to(torch.bfloat16).cuda()
    w_uv = wkv_b.view(H, 256, C)[:, 128:]
    y = ops.absorb_bmm_fp8(o, w_uv, sc, 4, 8, 0, 1).reshape(bs, H * 128)
    q_ref, s_ref = ops.act_quant_deepseek_v3(y.contiguous())
    q, s = ops.absorb_uv_quant_fp8(o, w_uv, sc, 4, 8, 1)
    assert np.array_equal(bits8(q), bits8(q_ref)) and torch.equal(s, s_ref)


@pytest.mark.parametrize("rows,H", [(256, 16), (300, 16), (1031, 5), (2048, 16)])
def test_absorb_projections_over_prefill_rows_equal_the_decode_kernels(rows, H):
    """>= 256 rows take the prefill forms (absorb.hip: weights de-quantised once per 8 token tiles); the same rows in chunks
    below that size take the decode kernels: bit-identical outputs for the W_UK absorb (K = 128, N = 512), the W_UV projection
    as plain bmm (K = 512: stays on the decode kernel) and the W_UV projection + act_quant (codes and scales)."""
    from chitu_amd import ops

    C = 512
    g = torch.Generator().manual_seed(rows + H)
    wkv_b = (torch.randn(H * 256, C, generator=g) * 0.5).to(torch.float8_e4m3fn)
    sc = (torch.rand(H * 2, C // 128, generator=g) * 0.02 + 0.01).cuda()
    w_uk_t = wkv_b.view(torch.uint8).view(H, 256, C)[:, :128].transpose(1, 2).contiguous().view(torch.float8_e4m3fn).cuda()
    w_uv = wkv_b.cuda().view(H, 256, C)[:, 128:]
    q_nope = torch.randn(rows, H, 192, generator=g).to(torch.bfloat16).cuda()[..., :128]
    o = torch.randn(rows, H, C, generator=g).to(torch.bfloat16).cuda()
    chunks = [(a, min(a + 200, rows)) for a in range(0, rows, 200)]
    full = ops.absorb_bmm_fp8(q_nope, w_uk_t, sc, 0, 8, 1, 0)
    parts = torch.cat([ops.absorb_bmm_fp8(q_nope[a:b], w_uk_t, sc, 0, 8, 1, 0) for a, b in chunks])
    assert tuple(full.shape) == (rows, H, C) and np.array_equal(bits16(full), bits16(parts))
    qf, sf = ops.absorb_uv_quant_fp8(o, w_uv, sc, 4, 8, 1)
    qp, sp = zip(*[ops.absorb_uv_quant_fp8(o[a:b], w_uv, sc, 4, 8, 1) for a, b in chunks])
    assert np.array_equal(bits8(qf), bits8(torch.cat(qp))) and torch.equal(sf, torch.cat(sp))
    assert torch.equal(full, ops.absorb_bmm_fp8(q_nope, w_uk_t, sc, 0, 8, 1, 0))


@pytest.mark.parametrize("bs,lens,splits", [(1, [1000], 17), (16, [1024] * 16, 16), (5, [1, 64, 65, 700, 130], 2), (3, [10, 20, 30], 4)])
def test_merge_folded_into_uv_projection_is_bit_identical(bs, lens, splits):
    """chitu_hip_mla_decode(out=NULL) + chitu_hip_mla_merge_absorb_uv_quant_fp8 ==
    chitu_hip_mla_decode (with its merge pass) + chitu_hip_absorb_uv_quant_fp8, including splits
    that see no token (lse = -inf)."""
    from chitu_amd import ops
    from chitu_amd.attn_backend import HipAttnBackend
    from tests.test_gpu_mla import make_case

    H, C = 16, 512
    q_nope, q_pe, cache, table, lens_t = make_case(bs, H, lens, pages=sum((l + 63) // 64 for l in lens) + 2, seed=bs)
    g = torch.Generator().manual_seed(bs + 90)
    wkv_b = (torch.randn(H * 256, C, generator=g) * 0.5).to(torch.float8_e4m3fn).cuda()
    sc = (torch.rand(H * 2, C // 128, generator=g) * 0.02 + 0.01).cuda()
    w_uv = wkv_b.view(H, 256, C)[:, 128:]
    be = HipAttnBackend(local_n_heads=H)
    args = (q_nope.cuda(), q_pe.cuda(), cache.cuda(), lens_t.cuda(), table.cuda(), 0.1)
    o = be.mla_decode(*args, num_splits=splits)
    q_ref, s_ref = ops.absorb_uv_quant_fp8(o, w_uv, sc, 4, 8, 1)
    part = be.mla_decode(*args, num_splits=splits, return_partials=True)
    assert isinstance(part, tuple) and part[1] == splits
    q, s = ops.mla_merge_absorb_uv_quant_fp8(part[0], splits, bs, w_uv, sc, 4, 8, 1)
    assert np.array_equal(bits8(q), bits8(q_ref)) and torch.equal(s, s_ref)


@pytest.mark.parametrize("bs", [1, 5, 16, 21])
def test_qkv_post_and_absorb_rope_equal_the_separate_launches(bs):
    """mla_qkv_post + absorb_bmm_rope_fp8 == rms_norm(quant) + mla_kv_prep + absorb_bmm_fp8, bit for bit
    (cache pages, fp8 q_a, scales, rotated q_pe, absorbed q_nope)."""
    from chitu_amd import ops

    g = torch.Generator().manual_seed(50 + bs)
    H, C = 16, 512
    q_a_kv = torch.randn(bs, 1536 + 576, generator=g).to(torch.bfloat16).cuda()
    q = torch.randn(bs, H, 192, generator=g).to(torch.bfloat16).cuda()
    cos, sin = torch.randn(bs, 32, generator=g).cuda(), torch.randn(bs, 32, generator=g).cuda()
    wq = (torch.rand(1536, generator=g) + 0.5).to(torch.bfloat16).cuda()
    wn = (torch.rand(512, generator=g) + 0.5).to(torch.bfloat16).cuda()
    pages = 2 * bs + 2
    cache = torch.randn(pages, 64, 576, generator=g).to(torch.bfloat16).cuda()
    table = torch.stack([torch.randperm(pages, generator=g)[:2] for _ in range(bs)]).to(torch.int32).cuda()
    lens = torch.tensor([(37 * i + (63 if i % 2 else 64)) % 128 for i in range(bs)], dtype=torch.int32).cuda()
    w_uk_t = (torch.randn(H, C, 128, generator=g) * 0.5).to(torch.float8_e4m3fn).cuda()
    sc = (torch.rand(H * 2, C // 128, generator=g) * 0.02 + 0.01).cuda()
    # separate launches
    _, qq_ref, qs_ref = ops.rms_norm(q_a_kv[:, :1536], wq, 1e-6, out_bf16=False, quant="act")
    q1, cache1 = q.clone(), cache.clone()
    ops.mla_kv_prep(q_a_kv[:, 1536:], q1[..., 128:], cos, sin, wn, 1e-6, cache1, table, lens)
    abs_ref = ops.absorb_bmm_fp8(q1[..., :128], w_uk_t, sc, 0, 8, 1, 0)
    # fused pair
    q2, cache2 = q.clone(), cache.clone()
    qq, qs = ops.mla_qkv_post(q_a_kv, 1536, wq, 1e-6, wn, 1e-6, cos, sin, cache2, table, lens)
    ab = ops.absorb_bmm_rope_fp8(q2[..., :128], w_uk_t, sc, 0, 8, 1, 0, q2[..., 128:], cos, sin)
    assert np.array_equal(bits8(qq), bits8(qq_ref)) and torch.equal(qs, qs_ref)
    assert torch.equal(cache2, cache1)
    assert torch.equal(q2, q1) and torch.equal(ab, abs_ref)


@pytest.mark.parametrize("bs,H", [(1, 16), (5, 16), (16, 16), (21, 4), (33, 2)])
def test_absorb_rope_kv_equals_the_separate_launches(bs, H):
    """absorb_bmm_rope_kv_fp8 (DeepSeek-V2-Lite's decode: no q low-rank path) == mla_kv_prep + absorb_bmm_fp8, bit for
    bit: cache pages (rows of other tokens untouched), rotated q_pe, absorbed q_nope; also with fewer heads than the
    16 tokens of a tile (a rider block then writes several rows)."""
    from chitu_amd import ops

    g = torch.Generator().manual_seed(150 + bs + H)
    C = 512
    q_kv = torch.randn(bs, H * 192 + 576, generator=g).to(torch.bfloat16).cuda()
    cos, sin = torch.randn(bs, 32, generator=g).cuda(), torch.randn(bs, 32, generator=g).cuda()
    wn = (torch.rand(512, generator=g) + 0.5).to(torch.bfloat16).cuda()
    pages = 2 * bs + 2
    cache = torch.randn(pages, 64, 576, generator=g).to(torch.bfloat16).cuda()
    table = torch.stack([torch.randperm(pages, generator=g)[:2] for _ in range(bs)]).to(torch.int32).cuda()
    lens = torch.tensor([(37 * i + (63 if i % 2 else 64)) % 128 for i in range(bs)], dtype=torch.int32).cuda()
    w_uk_t = (torch.randn(H, C, 128, generator=g) * 0.5).to(torch.float8_e4m3fn).cuda()
    sc = (torch.rand(H * 2, C // 128, generator=g) * 0.02 + 0.01).cuda()
    nq = H * 192
    qkv1, cache1 = q_kv.clone(), cache.clone()
    q1 = qkv1[:, :nq].view(bs, H, 192)
    ops.mla_kv_prep(qkv1[:, nq:], q1[..., 128:], cos, sin, wn, 1e-6, cache1, table, lens)
    abs_ref = ops.absorb_bmm_fp8(q1[..., :128], w_uk_t, sc, 0, 8, 1, 0)
    qkv2, cache2 = q_kv.clone(), cache.clone()
    q2 = qkv2[:, :nq].view(bs, H, 192)
    ab = ops.absorb_bmm_rope_kv_fp8(q2[..., :128], w_uk_t, sc, 0, 8, 1, 0, q2[..., 128:], cos, sin, qkv2[:, nq:], wn, 1e-6,
                                    cache2, table, lens)
    assert torch.equal(cache2, cache1) and not torch.equal(cache2, cache)
    assert torch.equal(qkv2, qkv1) and torch.equal(ab, abs_ref)


@pytest.mark.parametrize("bs,S", [(1, 2), (16, 2), (5, 3)])
def test_qkv_post_reads_split_k_planes_like_the_rounded_sum(bs, S):
    """mla_qkv_post(fp32 planes [S, bs, 2112]) == mla_qkv_post(bf16(plane 0 + plane 1 + ...)), bit for bit: the
    cross-workgroup K split of wqkv_a changes where the partial sums are added, not what its consumers compute."""
    from chitu_amd import ops

    g = torch.Generator().manual_seed(90 + bs + S)
    planes = torch.randn(S, bs, 1536 + 576, generator=g).cuda()
    total = planes[0].clone()
    for s in range(1, S):
        total += planes[s]
    q_a_kv = total.to(torch.bfloat16)
    cos, sin = torch.randn(bs, 32, generator=g).cuda(), torch.randn(bs, 32, generator=g).cuda()
    wq = (torch.rand(1536, generator=g) + 0.5).to(torch.bfloat16).cuda()
    wn = (torch.rand(512, generator=g) + 0.5).to(torch.bfloat16).cuda()
    pages = 2 * bs + 2
    cache = torch.randn(pages, 64, 576, generator=g).to(torch.bfloat16).cuda()
    table = torch.stack([torch.randperm(pages, generator=g)[:2] for _ in range(bs)]).to(torch.int32).cuda()
    lens = torch.tensor([(37 * i + (63 if i % 2 else 64)) % 128 for i in range(bs)], dtype=torch.int32).cuda()
    c1, c2 = cache.clone(), cache.clone()
    q1, s1 = ops.mla_qkv_post(q_a_kv, 1536, wq, 1e-6, wn, 1e-6, cos, sin, c1, table, lens)
    q2, s2 = ops.mla_qkv_post(planes, 1536, wq, 1e-6, wn, 1e-6, cos, sin, c2, table, lens)
    assert np.array_equal(bits8(q1), bits8(q2)) and torch.equal(s1, s2) and torch.equal(c1, c2)
    assert not torch.equal(c1, cache)


def _q_proj_case(bs, ql, seed):
    g = torch.Generator().manual_seed(seed)
    q_a_kv = (torch.randn(bs, ql + 576, generator=g) * (0.2 + torch.rand(bs, 1, generator=g) * 3)).to(torch.bfloat16).cuda()
    cos, sin = torch.randn(bs, 32, generator=g).cuda(), torch.randn(bs, 32, generator=g).cuda()
    wq = (torch.rand(ql, generator=g) + 0.5).to(torch.bfloat16).cuda()
    wn = (torch.rand(512, generator=g) + 0.5).to(torch.bfloat16).cuda()
    pages = 2 * bs + 2
    cache = torch.randn(pages, 64, 576, generator=g).to(torch.bfloat16).cuda()
    table = torch.randperm(pages, generator=g)[: 2 * bs].view(bs, 2).to(torch.int32).cuda()  # no page shared by two sequences
    lens = torch.tensor([(37 * i + (63 if i % 2 else 64)) % 128 for i in range(bs)], dtype=torch.int32).cuda()
    return g, q_a_kv, cos, sin, wq, wn, cache, table, lens


@pytest.mark.parametrize("bs", [1, 5, 16, 17, 32])
@pytest.mark.parametrize("ql,N", [(1536, 3072), (512, 200), (2048, 1536), (128, 64)])
def test_q_proj_one_launch_vs_the_separate_launches_and_the_oracle(bs, ql, N):
    """mla_q_proj (q_norm + act_quant as the wq_b GEMM's prologue, KV append on extra workgroups) against
    mla_qkv_post + fp8_gemm_deepseek_v3: the page rows bit for bit (same code); q (fp32) within 2e-3 of the pair's peak (the
    mean square is summed in another order: the last bit of a few bf16 norms, hence one fp8 step on a few inputs) and
    within 1e-2 of the CPU oracle's rms_norm -> act_quant -> fp8 GEMM."""
    from chitu_amd import ops
    from oracle import fp8 as ofp8

    g, q_a_kv, cos, sin, wq, wn, cache, table, lens = _q_proj_case(bs, ql, 300 + bs + ql)
    w = (torch.randn(N, ql, generator=g) * 0.5).to(torch.float8_e4m3fn).cuda()
    ws = (torch.rand((N + 127) // 128, ql // 128, generator=g) * 0.02 + 0.01).cuda()
    c1, c2 = cache.clone(), cache.clone()
    qq, qs = ops.mla_qkv_post(q_a_kv, ql, wq, 1e-6, wn, 1e-6, cos, sin, c1, table, lens)
    ref = ops.fp8_gemm_deepseek_v3(qq, qs, w, ws, out_dtype=torch.float32)
    out = ops.mla_q_proj(q_a_kv, ql, wq, 1e-6, w, ws, wn, 1e-6, cos, sin, c2, table, lens, out_dtype=torch.float32)
    torch.cuda.synchronize()
    bad = (c1 != c2).nonzero()
    assert bad.shape[0] == 0, (bad.shape[0], bad[:8].tolist(), table.tolist(), lens.tolist(),
                               [(c1[tuple(i)].item(), c2[tuple(i)].item(), cache[tuple(i)].item()) for i in bad[:8].tolist()])
    assert not torch.equal(c2, cache)
    assert_close(out, ref, 2e-3)# fp32 outputs: a bf16 output's own last bit would be 4e-3 of the peak
    out = ops.mla_q_proj(q_a_kv, ql, wq, 1e-6, w, ws, wn, 1e-6, cos, sin, c2, table, lens, out_dtype=torch.bfloat16)
    y = torch.nn.functional.rms_norm(q_a_kv[:, :ql].cpu().float(), (ql,), wq.cpu().float(), 1e-6).to(torch.bfloat16)
    oq, os_ = ofp8.act_quant_deepseek_v3(y)
    o_ref = ofp8.fp8_gemm_deepseek_v3(oq, os_, w.cpu(), ws.cpu(), torch.bfloat16)
    assert_close(out, o_ref, 1e-2)


@pytest.mark.parametrize("bs", [1, 16, 23])
def test_q_proj_prologue_quantises_like_act_quant(bs):
    """The prologue seen through an identity wq_b (fp8 1.0 on the diagonal, scales 1.0, fp32 output): what comes out is
    fp8(y / s) * s of the normalised row, i.e. the dequantised act_quant -- equal to the unfused launch's (q, s) on all
    but a few elements (those whose bf16 norm sits on a rounding boundary of the mean square's last bit), and never
    further away than one fp8 step."""
    from chitu_amd import ops

    ql = 1536
    g, q_a_kv, cos, sin, wq, wn, cache, table, lens = _q_proj_case(bs, ql, 700 + bs)
    w = torch.eye(ql).to(torch.float8_e4m3fn).cuda()
    ws = torch.ones(ql // 128, ql // 128).cuda()
    c1, c2 = cache.clone(), cache.clone()
    qq, qs = ops.mla_qkv_post(q_a_kv, ql, wq, 1e-6, wn, 1e-6, cos, sin, c1, table, lens)
    deq = qq.float().view(bs, ql // 128, 128) * qs[:, :, None]
    out = ops.mla_q_proj(q_a_kv, ql, wq, 1e-6, w, ws, wn, 1e-6, cos, sin, c2, table, lens, out_dtype=torch.float32)
    out = out.view(bs, ql // 128, 128)
    diff = (out != deq)
    assert diff.float().mean().item() < 0.01, diff.float().mean().item()
    # one e4m3 step is at most 2^-3 of the value's binade: <= 1/8 * |v| (+ the scale's own last-bit change)
    assert ((out - deq).abs() <= 0.13 * deq.abs().clamp_min(1e-30) + 1e-6 * qs[:, :, None]).all()
    assert torch.equal(c1, c2)


def test_q_proj_unsupported_shapes_say_so():
    from chitu_amd import ops

    assert not ops.mla_q_proj_fits(33, 1536) and not ops.mla_q_proj_fits(16, 2176) and not ops.mla_q_proj_fits(0, 1536)
    g, q_a_kv, cos, sin, wq, wn, cache, table, lens = _q_proj_case(33, 1536, 9)
    w = torch.zeros(64, 1536).to(torch.float8_e4m3fn).cuda()
    with pytest.raises(AssertionError):
        ops.mla_q_proj(q_a_kv, 1536, wq, 1e-6, w, torch.ones(1, 12).cuda(), wn, 1e-6, cos, sin, cache, table, lens)


@pytest.mark.parametrize("E,groups,topk,S,bias", [(256, (8, 4), 8, 16, True), (256, (8, 4), 8, 0, True), (256, (4, 2), 8, 3, True),
                                                  (256, (8, 4), 8, 16, False), (128, (1, 1), 6, 5, True), (64, (2, 1), 4, 0, True)])
def test_gate_route_fast_path_equals_generic_kernel(E, groups, topk, S, bias, monkeypatch):
    """The key-based routing kernel and the generic rank-loop kernel are the same function, on
    tie-heavy inputs too (coarse logits => many equal bf16 scores; the tie rule is lower index first)."""
    from chitu_amd import _lib
    from chitu_amd._lib import f32, i32, i64, ptr, stream_ptr

    M = 37
    g = torch.Generator().manual_seed(E + S + topk)
    for coarse in (False, True):
        if S == 0:
            logits = torch.randn(M, E, generator=g) * 2
            logits = ((logits * 2).round() / 2 if coarse else logits).to(torch.bfloat16).cuda()
        else:
            part = torch.randn(S, M, E, generator=g)
            logits = ((part * 4).round() / 4 if coarse else part).float().cuda()
        b = ((torch.randn(E, generator=g) * (0.25 if coarse else 0.05)).to(torch.bfloat16).cuda()) if bias else None
        outs = []
        for slow in (0, 1):
            w = torch.zeros(M, topk + 2, dtype=torch.bfloat16, device="cuda")
            ids = torch.zeros(M, topk + 2, dtype=torch.int64, device="cuda")
            with _lib.debug_option("gate_generic", slow):
                rc = _lib.lib().chitu_hip_gate_route(ptr(logits), i32(S), i64(M), i32(E), ptr(b), i32(groups[0]), i32(groups[1]),
                                                     i32(topk), i32(1), f32(2.5), ptr(w), ptr(ids), i32(topk + 2), i32(E),
                                                     f32(1.0), i32(2), stream_ptr())
            assert rc == 0
            torch.cuda.synchronize()
            outs.append((w.cpu(), ids.cpu()))
        assert torch.equal(outs[0][1], outs[1][1])
        assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16))


@pytest.mark.parametrize("rows,terms,dim,quant", [(16, 9, 7168, "act"), (1, 9, 7168, "act"), (5, 3, 2048, "group"), (7, 16, 512, None), (3, 2, 8192, "act")])
def test_topk_sum_folded_into_rmsnorm_is_bit_identical(rows, terms, dim, quant):
    """rms_norm(add=[rows, terms, dim]) == chitu_hip_moe_sum followed by rms_norm(add=[rows, dim])."""
    from chitu_amd import _lib, ops
    from chitu_amd._lib import i32, i64, ptr, stream_ptr

    g = torch.Generator().manual_seed(rows * 31 + terms)
    x = torch.randn(rows, dim, generator=g).to(torch.bfloat16).cuda()
    c3 = (torch.randn(rows, terms, dim, generator=g) * 0.3).to(torch.bfloat16).cuda()
    w = (torch.rand(dim, generator=g) + 0.5).to(torch.bfloat16).cuda()
    summed = torch.empty(rows, dim, dtype=torch.bfloat16, device="cuda")
    assert _lib.lib().chitu_hip_moe_sum(ptr(c3), ptr(summed), i64(rows), i32(terms), i64(dim), stream_ptr()) == 0
    ref = ops.rms_norm(x, w, 1e-6, quant=quant, add=summed)
    got = ops.rms_norm(x, w, 1e-6, quant=quant, add=c3)
    assert len(ref) == len(got)
    for a, b in zip(ref, got):
        if a.dtype == torch.float8_e4m3fn:
            assert np.array_equal(bits8(a), bits8(b))
        else:
            assert torch.equal(a, b)


@pytest.mark.parametrize("make_args", [tiny_args, v2lite_like_args], ids=["v3_like", "v2lite_like"])
def test_prefill_equals_token_by_token_decode(make_args):
    """Size-independent property: running a prompt through prefill (ragged batch, causal attention over
    the prompt, page writes by the cache manager) gives the same last-token logits and the same KV pages
    as feeding the same tokens one at a time through the decode step -- and generation continues from
    either state identically (within the fp8/bf16 noise of different split-K orders)."""
    args = make_args()
    model, cache = build(args, max_reqs=4, max_seq=512)
    g = torch.Generator().manual_seed(3)
    prompts = [torch.randint(0, args.vocab_size, (n,), generator=g).tolist() for n in (1, 70, 9)]

    # (a) token by token through decode
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
        for k, i in enumerate(live):
            if step == len(prompts[i]) - 1:
                last_logits[i] = logits[k].clone()
    kv_decode = {}
    for i, r in enumerate(reqs):
        rows = torch.cat([cache.paged_kv_cache[:, b] for b in cache.block_table[r]], dim=1)[:, : len(prompts[i])]
        kv_decode[i] = rows.clone()
        cache.finalize_cache_all_decode(r)

    # (b) one prefill call
    preqs = ["p0", "p1", "p2"]
    logits_p = model.prefill(prompts, preqs)
    assert tuple(logits_p.shape) == (3, args.vocab_size) and logits_p.dtype == torch.float32
    for i, r in enumerate(preqs):
        assert cache.seq_lens[r] == len(prompts[i])
        rows = torch.cat([cache.paged_kv_cache[:, b] for b in cache.block_table[r]], dim=1)[:, : len(prompts[i])]
        # layer 0's rows see identical arithmetic (embedding, norm, one GEMM row by row); deeper layers carry
        # the fp8 re-quantisation noise of different split-K / KV-split summation orders
        assert_close(rows[0], kv_decode[i][0], 5e-3)
        assert_close(rows, kv_decode[i], 6e-2)
        assert_close(logits_p[i], last_logits[i], 6e-2, what=i)
    # generation continues from the prefilled state
    tok = logits_p.argmax(-1)
    cache.prepare_cache_decode(preqs)
    cache.prepare_block_table_for_decode(preqs)
    nxt = model.decode(tok, use_graph=True)
    assert torch.isfinite(nxt).all()
    cache.finalize_cache_single_decode(preqs)
    for r in preqs:
        cache.finalize_cache_all_decode(r)
    # end-to-end helper: deterministic, right shape, pages returned
    free_before = len(cache.free_blocks)
    out1 = model.generate(prompts, 4)
    out2 = model.generate(prompts, 4)
    assert tuple(out1.shape) == (3, 4) and torch.equal(out1, out2) and len(cache.free_blocks) == free_before


def test_long_decode_run_graph_equals_eager_across_page_boundaries():
    """70 consecutive decode steps (contexts cross two 64-token page boundaries, split ranges and tile
    counts change under a fixed captured graph): hipGraph replay and eager launches stay bit-identical in
    logits and in the KV pages they write, with the token stream teacher-forced from the graph run."""
    args = tiny_args()
    model, cache = build(args, max_reqs=4, max_seq=512)
    starts = (60, 63, 127)

    def fresh(tag):
        reqs = [f"{tag}{i}" for i in range(3)]
        g = torch.Generator().manual_seed(99)
        for r, n in zip(reqs, starts):
            cache.register_sequence(r, n)
            rows = (torch.randn(args.n_layers, 256, 576, generator=g) * 0.5).to(torch.bfloat16).cuda()
            for p, blk in enumerate(cache.block_table[r]):
                cache.paged_kv_cache[:, blk] = rows[:, p * 64 : (p + 1) * 64]
        return reqs

    def gather(reqs):
        return [torch.cat([cache.paged_kv_cache[:, b] for b in cache.block_table[r]], dim=1)[:, : cache.seq_lens[r]].clone()
                for r in reqs]

    def run(reqs, use_graph, forced=None):
        toks = torch.tensor([5, 17, 900], dtype=torch.int64, device="cuda")
        logits_all, toks_all = [], []
        for step in range(70):
            cache.prepare_cache_decode(reqs)
            cache.prepare_block_table_for_decode(reqs)
            logits = model.decode(toks, use_graph=use_graph).clone()
            cache.finalize_cache_single_decode(reqs)
            logits_all.append(logits)
            toks = logits.argmax(-1) if forced is None else forced[step]
            toks_all.append(toks.clone())
        return logits_all, toks_all

    ra = fresh("ga")
    lg, tg = run(ra, True)
    kv_g = gather(ra)
    for r in ra:
        cache.finalize_cache_all_decode(r)
    rb = fresh("ea")
    le, _ = run(rb, False, forced=tg)
    kv_e = gather(rb)
    for step, (a, b) in enumerate(zip(lg, le)):
        assert torch.equal(a, b), step
    for a, b in zip(kv_g, kv_e):
        assert torch.equal(a, b)
    assert all(torch.isfinite(x).all() for x in lg) and cache.seq_lens[rb[0]] == starts[0] + 70


def test_large_batch_matches_small_batches():
    """Batch invariance: one decode step over 70 ragged sequences (multi-tile GEMM paths, > 32 sequences
    => un-fused split merge, 350 routed slots) gives every sequence the logits it gets when decoded in
    batches of 7 (different K-split heuristics => fp8-noise tolerance), graph replay == eager."""
    args = tiny_args()
    model, cache = build(args, max_reqs=72, max_seq=320)
    g = torch.Generator().manual_seed(17)
    n = 70
    lens = torch.randint(1, 200, (n,), generator=g).tolist()
    toks = torch.randint(0, args.vocab_size, (n,), generator=g)

    def fill(tag):
        reqs = [f"{tag}{i}" for i in range(n)]
        gg = torch.Generator().manual_seed(5)
        for r, L in zip(reqs, lens):
            cache.register_sequence(r, L)
            rows = (torch.randn(args.n_layers, 256, 576, generator=gg) * 0.5).to(torch.bfloat16).cuda()
            for p, blk in enumerate(cache.block_table[r]):
                cache.paged_kv_cache[:, blk] = rows[:, p * 64 : (p + 1) * 64]
        return reqs

    def step(reqs, idx, use_graph):
        ids = [reqs[i] for i in idx]
        cache.prepare_cache_decode(ids)
        cache.prepare_block_table_for_decode(ids)
        out = model.decode(toks[idx].cuda(), use_graph=use_graph).clone()
        cache.finalize_cache_single_decode(ids)
        return out

    big = fill("b")
    snap = cache.paged_kv_cache.clone()
    full = step(big, list(range(n)), use_graph=False)
    assert torch.isfinite(full).all() and tuple(full.shape) == (n, args.vocab_size)
    # graph replay of the same step on the restored cache
    cache.paged_kv_cache.copy_(snap)
    for r, L in zip(big, lens):
        cache.seq_lens[r] = L
    assert torch.equal(step(big, list(range(n)), use_graph=True), full)
    for r in big:
        cache.finalize_cache_all_decode(r)
    small = fill("s")
    for s0 in range(0, n, 7):
        idx = list(range(s0, s0 + 7))
        part = step(small, idx, use_graph=False)
        assert_close(part, full[idx], 4e-2, what=s0)


@pytest.mark.parametrize("make_args", [tiny_args, v2lite_like_args], ids=["v3_like", "v2lite_like"])
def test_tile_major_activations_leave_the_decode_step_bit_identical(make_args, monkeypatch):
    """The fused step keeps the fp8 activations of wqkv_a / wo / the dense MLP tile-major between its own launches
    (ops.TiledQuant).  Same codes, same scales, same GEMM arithmetic: the logits and the KV rows of several decode steps
    are bit-identical with the layout switched off (ops._TILE_MAJOR = False)."""
    from chitu_amd import ops

    args = make_args()
    model, cache = build(args, max_reqs=4, max_seq=512)

    def run(tag, flag):
        monkeypatch.setattr(ops, "_TILE_MAJOR", flag)
        reqs = [f"{tag}{i}" for i in range(3)]
        g = torch.Generator().manual_seed(5)
        for r, n in zip(reqs, (60, 3, 127)):
            cache.register_sequence(r, n)
            rows = (torch.randn(args.n_layers, 256, 576, generator=g) * 0.5).to(torch.bfloat16).cuda()
            for p, blk in enumerate(cache.block_table[r]):
                cache.paged_kv_cache[:, blk] = rows[:, p * 64 : (p + 1) * 64]
        toks = torch.tensor([5, 17, 900], dtype=torch.int64, device="cuda")
        out = []
        for _ in range(6):
            cache.prepare_cache_decode(reqs)
            cache.prepare_block_table_for_decode(reqs)
            logits = model.decode(toks, use_graph=False).clone()
            cache.finalize_cache_single_decode(reqs)
            out.append(logits)
            toks = logits.argmax(-1)
        kv = [torch.cat([cache.paged_kv_cache[:, b] for b in cache.block_table[r]], dim=1)[:, : cache.seq_lens[r]].clone() for r in reqs]
        for r in reqs:
            cache.finalize_cache_all_decode(r)
        return out, kv

    (la, ka), (lb, kb) = run("t", True), run("r", False)
    for a, b in zip(la, lb):
        assert torch.equal(a, b)
    for a, b in zip(ka, kb):
        assert torch.equal(a, b)


# ---------------------------------------------------------------- attn_norm as the prologue of the first projection
@pytest.mark.parametrize("M,N,K,terms", [(1, 2112, 7168, 9), (1, 2112, 7168, 1), (2, 2112, 7168, 1), (1, 3648, 2048, 8),
                                         (1, 4608, 7168, 1), (2, 4608, 7168, 1), (1, 832, 512, 5), (1, 2112, 7168, 16),
                                         (1, 1000, 1024, 3)])
def test_attn_norm_in_the_first_projection_prologue_is_bit_identical(M, N, K, terms):
    """ops.fp8_linear_add_norm (ONE launch) against rms_norm(add=..., quant="act") + fp8_gemm_deepseek_v3 on row-major
    and on tile-major activations: the residual stream and the projection's output are identical bit for bit."""
    from chitu_amd import ops

    if not ops.fp8_linear_add_norm_fits(M, N, K, terms):
        pytest.skip("shape outside the fused launch")
    g = torch.Generator().manual_seed(M * 1000 + N + K + terms)
    x = (torch.randn(M, K, generator=g) * 2).to(torch.bfloat16).cuda()
    add = (torch.randn(*((M, terms, K) if terms > 1 else (M, K)), generator=g)).to(torch.bfloat16).cuda()
    if K >= 256:  # an all-zero 128-group: act_quant's 0 / 0 (NaN codes and scale 0 in both forms)
        x[0, 128:256] = 0
        add[0, ..., 128:256] = 0
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.5).to(torch.float8_e4m3fn).cuda()
    ws = (torch.rand((N + 127) // 128, K // 128, generator=g) * 0.02 + 0.01).cuda()
    x_ref, _, q, s = ops.rms_norm(x, nw, 1e-6, out_bf16=False, quant="act", add=add)
    out_ref = ops.fp8_gemm_deepseek_v3(q, s, w, ws, out_dtype=torch.bfloat16)
    x_new, out = ops.fp8_linear_add_norm(x, add, nw, 1e-6, w, ws)
    assert torch.equal(x_new, x_ref)
    assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16))  # bitwise (NaN rows included)
    if ops.tile_major_ok(M):
        _, _, tq, _ = ops.rms_norm(x, nw, 1e-6, out_bf16=False, quant="act", add=add, tile_major=True)
        out_tm = ops.fp8_gemm_deepseek_v3(tq, None, w, ws, out_dtype=torch.bfloat16)
        assert torch.equal(out.view(torch.int16), out_tm.view(torch.int16))
    out32 = ops.fp8_linear_add_norm(x, add, nw, 1e-6, w, ws, out_dtype=torch.float32)[1]
    assert torch.equal(out32.view(torch.int32), ops.fp8_gemm_deepseek_v3(q, s, w, ws, out_dtype=torch.float32).view(torch.int32))


@pytest.mark.parametrize("bs", [1, 2])
def test_decode_step_with_norms_in_the_gemm_prologues_equals_the_step_without(bs, monkeypatch):
    """A model wide enough for the fused launch (dim 4096), three decode steps eagerly and through the graph, with attn_norm
    inside the first projection switched on against off: identical logits and KV rows."""
    from chitu_amd import deepseek_v3 as ds
    from chitu_amd.deepseek_v3 import DeepSeekV3Args

    args = DeepSeekV3Args(vocab_size=1024, dim=4096, inter_dim=1024, moe_inter_dim=256, n_layers=3, n_dense_layers=1, n_heads=16,
                          n_routed_experts=32, n_shared_experts=1, n_activated_experts=4, n_expert_groups=4, n_limited_groups=2,
                          q_lora_rank=256, gate_bias=True)
    model, cache = build(args, max_reqs=2, max_seq=256)
    taken = {"attn": 0}
    real_a = ds.ops.fp8_linear_add_norm
    monkeypatch.setattr(ds.ops, "fp8_linear_add_norm", lambda *a, **k: (taken.__setitem__("attn", taken["attn"] + 1), real_a(*a, **k))[1])

    def run(tag, attn_limit, use_graph):
        monkeypatch.setattr(ds, "FUSE_ATTN_NORM_MAX_BS", attn_limit)
        model.graphs.clear()
        reqs = [f"{tag}{i}" for i in range(bs)]
        g = torch.Generator().manual_seed(5)
        for r, n in zip(reqs, (60, 3)):
            cache.register_sequence(r, n)
            rows = (torch.randn(args.n_layers, 64, 576, generator=g) * 0.5).to(torch.bfloat16).cuda()
            for blk in cache.block_table[r]:
                cache.paged_kv_cache[:, blk] = rows
        toks = torch.tensor([5, 17][:bs], dtype=torch.int64, device="cuda")
        out = []
        for _ in range(3):
            cache.prepare_cache_decode(reqs)
            cache.prepare_block_table_for_decode(reqs)
            logits = model.decode(toks, use_graph=use_graph).clone()
            cache.finalize_cache_single_decode(reqs)
            out.append(logits)
            toks = logits.argmax(-1)
        kv = [torch.cat([cache.paged_kv_cache[:, b] for b in cache.block_table[r]], dim=1)[:, : cache.seq_lens[r]].clone() for r in reqs]
        for r in reqs:
            cache.finalize_cache_all_decode(r)
        return out, kv

    base = run("a", 0, False)
    assert taken == {"attn": 0}
    for i, (attn_limit, use_graph) in enumerate([(2, False), (2, True)]):
        before = dict(taken)
        got = run(f"v{i}", attn_limit, use_graph)
        # attn_norm fusion: one row with the experts' terms, two rows only behind a plain add (the dense layer)
        assert (taken["attn"] > before["attn"]) == (attn_limit > 0), "attn_norm fusion taken / not taken as asked"
        for a, b in zip(base[0] + base[1], got[0] + got[1]):
            assert torch.equal(a, b), (attn_limit, use_graph)
