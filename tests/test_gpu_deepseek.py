"""End-to-end decode-layer parity: chitu_amd.deepseek_v3 (HIP ops) vs oracle.deepseek (CPU restatement
of the reference's model_deepseek_v3.py decode path), plus RMSNorm / absorb op tests and hipGraph replay."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import deepseek as ods
from oracle import fp8 as ofp8
from tests.util import bits16, bits8, max_rel_to_peak

pytestmark = pytest.mark.gpu


def tiny_args():
    from chitu_amd.deepseek_v3 import DeepSeekV3Args

    return DeepSeekV3Args(
        vocab_size=1024, dim=512, inter_dim=1024, moe_inter_dim=256, n_layers=3, n_dense_layers=1, n_heads=16,
        n_routed_experts=16, n_shared_experts=1, n_activated_experts=4, n_expert_groups=4, n_limited_groups=2,
        q_lora_rank=256, gate_bias=True,
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


def test_layerwise_parity_and_graph_replay():
    args = tiny_args()
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

            def hooked(inp, _orig=orig, _store=routing):
                w, idx = _orig(inp)
                _store["w"], _store["i"] = w.cpu(), idx.cpu()
                return w, idx

            layer.ffn.gate.forward = hooked
        with torch.inference_mode():
            y = layer(x.cuda(), cos.cuda(), sin.cuda()).cpu()
        if layer.is_moe:
            layer.ffn.gate.forward = orig
        rt = (routing["w"], routing["i"]) if layer.is_moe else None
        y_ref, new_cache, _ = ods.block(params, i, x, cos, sin, shadow[i], table, lens_excl, cfg, layer.is_moe, rt)
        # the appended KV row: same bf16 values up to one rounding of the fp8 GEMM/norm chain
        assert max_rel_to_peak(cache.paged_kv_cache[i].cpu(), new_cache) < 1e-2
        err = max_rel_to_peak(y, y_ref)
        worst = max(worst, err)
        assert err < 2e-2, (i, err)
        if layer.is_moe:
            # gate parity on the oracle's own normalised input: same experts except bf16 ties
            hn = ods.rms_norm(x + ods.attention_decode(params, f"layers.{i}.attn.", ods.rms_norm(x, params[f"layers.{i}.attn_norm.weight"], cfg["eps"]),
                                                       cos, sin, shadow[i], table, lens_excl, cfg)[0],
                              params[f"layers.{i}.ffn_norm.weight"], cfg["eps"])
            w_ref, i_ref = ods.gate(hn, params[f"layers.{i}.ffn.gate.weight"], params[f"layers.{i}.ffn.gate.bias"],
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
