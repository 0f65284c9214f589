"""Oracle for one DeepSeek-V3 decode layer / step (torch CPU).  TEST INFRASTRUCTURE ONLY.

Restates the per-token path of chitu/models/model_deepseek_v3.py with the reference's op order and
rounding points, on top of the op-level oracles (oracle.fp8 / kv / mla / moe):
  linear_deepseek_v3:53-106, AttentionDeepSeekV3._run_linear:475-536 (absorb-without-precomp, incl.
  the materialised bf16 weight_dequant of wkv_b :511-528), decode_forward_paged:672-699,
  MLPDeepSeekV3.forward:755-771, GateDeepSeekV3.forward:810-842, MoEDeepSeekV3.forward:921-1011,
  TransformerBlockDeepSeekV3.forward:1100-1114, RMSNorm.forward (models/model.py:50-78).
Weights are passed as a plain dict of CPU tensors (names = the chitu_amd.deepseek_v3 module tree).
"""

import torch
import torch.nn.functional as F

from . import fp8, kv, mla, moe

BLOCK = 128


def rms_norm(x, w, eps):
    """models/model.py:72-76 with compute_dtype = x.dtype."""
    return F.rms_norm(x, (x.shape[-1],), w, eps).to(x.dtype)


def gate(x, weight, bias, n_groups, topk_groups, topk, score_func, route_scale):
    """model_deepseek_v3.py:810-842 verbatim semantics (bf16 scores, torch.topk tie-breaking)."""
    return gate_from_logits(F.linear(x, weight), bias, n_groups, topk_groups, topk, score_func, route_scale)


def gate_from_logits(logits, bias, n_groups, topk_groups, topk, score_func, route_scale, return_masked=False):
    """Everything after `scores = linear(x, self.weight)` (model_deepseek_v3.py:821-842)."""
    x = logits
    scores = logits
    scores = scores.softmax(dim=-1, dtype=torch.float32) if score_func == "softmax" else scores.sigmoid()
    original = scores
    if bias is not None:
        scores = scores + bias
    pre_mask, group_scores = scores, None
    if n_groups > 1:
        scores = scores.view(x.size(0), n_groups, -1)
        group_scores = scores.amax(dim=-1) if bias is None else scores.topk(2, dim=-1)[0].sum(dim=-1)
        idx = group_scores.topk(topk_groups, dim=-1)[1]
        mask = torch.zeros_like(scores[..., 0]).scatter_(1, idx, True)
        scores = (scores * mask.unsqueeze(-1)).flatten(1)
    indices = torch.topk(scores, topk, dim=-1)[1]
    weights = original.gather(1, indices)
    if score_func == "sigmoid":
        weights = weights / weights.sum(dim=-1, keepdim=True)
    weights = weights * route_scale
    if return_masked:
        return weights.type_as(x), indices, dict(masked=scores, original=original, pre_mask=pre_mask,
                                                 group_scores=group_scores)
    return weights.type_as(x), indices


def attention_decode(p, pre, x, cos, sin, cache_layer, block_table, lens_excl, cfg):
    """Returns (wo output [bs, dim], updated cache layer)."""
    H, C, R, NOPE, V, QL = cfg["H"], cfg["C"], cfg["R"], cfg["NOPE"], cfg["V"], cfg["QL"]
    bs = x.shape[0]
    if QL > 0:
        q_a_kv = fp8.linear_deepseek_v3(x, p[pre + "wqkv_a.weight"], p[pre + "wqkv_a.scale"])
        q_a, kvr = q_a_kv[:, :QL], q_a_kv[:, QL:]
        q = fp8.linear_deepseek_v3(rms_norm(q_a, p[pre + "q_norm.weight"], cfg["eps"]), p[pre + "wq_b.weight"], p[pre + "wq_b.scale"])
    else:  # q_lora_rank == 0 (DeepSeek-V2-Lite): q = wq(x) (model_deepseek_v3.py:423-432), merged with wkv_a
        q_kv = fp8.linear_deepseek_v3(x, p[pre + "wq_kv_a.weight"], p[pre + "wq_kv_a.scale"])
        q, kvr = q_kv[:, : H * (NOPE + R)], q_kv[:, H * (NOPE + R) :]
    q = q.reshape(bs, H, NOPE + R)
    q_nope, q_pe = q[..., :NOPE], q[..., NOPE:]
    kv_c, k_pe = kvr[:, :C], kvr[:, C:]
    q_pe, k_pe = kv.apply_rotary_pos_emb(q_pe, k_pe, cos, sin, "llama")
    wkv_b = fp8.weight_dequant_deepseek_v3(p[pre + "wkv_b.weight"], p[pre + "wkv_b.scale"]).view(H, NOPE + V, C)
    q_abs = torch.einsum("shd,hdc->shc", q_nope, wkv_b[:, :NOPE])
    this_kv = rms_norm(kv_c, p[pre + "kv_norm.weight"], cfg["eps"])
    this_kv_pe = torch.cat([this_kv, k_pe], dim=-1)
    o, cache_layer = mla.mla_attn_with_kvcache(
        q_abs, q_pe, cache_layer, this_kv_pe.view(bs, 1, 1, -1), lens_excl, lens_excl + 1, block_table, cfg["scale"]
    )
    o = o.to(x.dtype).view(bs, 1, H, C)
    o = torch.einsum("bshc,hdc->bshd", o, wkv_b[:, -V:]).reshape(bs, H * V)
    return fp8.linear_deepseek_v3(o, p[pre + "wo.weight"], p[pre + "wo.scale"]), cache_layer


def mlp(p, pre, x):
    h = fp8.linear_deepseek_v3(x, p[pre + "w1w3.weight"], p[pre + "w1w3.scale"])
    d = h.shape[-1] // 2
    return fp8.linear_deepseek_v3(F.silu(h[..., :d]) * h[..., d:], p[pre + "w2.weight"], p[pre + "w2.scale"])


def moe_layer(p, pre, x, cfg, routing=None):
    if routing is None:
        routing = gate(x, p[pre + "gate.weight"], p.get(pre + "gate.bias"), cfg["n_groups"], cfg["topk_groups"],
                       cfg["topk"], cfg["score_func"], cfg["route_scale"])
    weights, indices = routing
    nr = cfg["n_routed"]
    w13, s13, w2, s2 = p[pre + "w1w3_weight"], p[pre + "w1w3_scale"], p[pre + "w2_weight"], p[pre + "w2_scale"]
    y = None
    for i in range(nr, w13.shape[0]):
        h = fp8.linear_deepseek_v3(x, w13[i], s13[i])
        d = h.shape[-1] // 2
        yi = fp8.linear_deepseek_v3(F.silu(h[..., :d]) * h[..., d:], w2[i], s2[i])
        y = yi if y is None else y + yi
    y1 = moe.fused_experts_fp8(x, w13[:nr], w2[:nr], weights, indices, s13[:nr], s2[:nr])
    return (y1 if y is None else y + y1), routing


def moe_layer_ep(p, pre, x, cfg, expert_map, routing=None):
    """One rank's partial sum of the expert-parallel MoE (the reference's hooks:
    model_deepseek_v3.py:870-880 moe_world_size / experts_start_idx, :1004 expert_map): the routed
    experts this rank holds at full width (`w1w3_weight [n_local, 2I, dim]`) over the GLOBAL routing,
    plus its 1/ep slice of the shared experts' width (`shared.w1w3`, `shared.w2`: a row/column-parallel
    MLP).  Summing the ranks' results (the layer's all-reduce) gives the layer output."""
    if routing is None:
        routing = gate(x, p[pre + "gate.weight"], p.get(pre + "gate.bias"), cfg["n_groups"], cfg["topk_groups"],
                       cfg["topk"], cfg["score_func"], cfg["route_scale"])
    weights, indices = routing
    y = moe.fused_experts_fp8(x, p[pre + "w1w3_weight"], p[pre + "w2_weight"], weights, indices,
                              p[pre + "w1w3_scale"], p[pre + "w2_scale"], expert_map=expert_map)
    if pre + "shared.w1w3.weight" in p:
        y = y + mlp(p, pre + "shared.", x)
    return y, routing


def moe_layer_loop(p, pre, x, cfg, routing=None):
    """The reference's OTHER MoE branch: the per-expert loop (model_deepseek_v3.py:1012-1061) it takes where the
    fused Triton kernel is unavailable.  Each expert's w2 output is rounded to bf16, scaled by its bf16 routing
    weight and accumulated into a bf16 `y` in expert order, then the shared experts are added.  Used to pin this
    file against the reference model's own CPU run (tests/golden/gen_ref_model.py), where that branch is the one
    that runs; the product path follows the fused branch (`moe_layer`)."""
    if routing is None:
        routing = gate(x, p[pre + "gate.weight"], p.get(pre + "gate.bias"), cfg["n_groups"], cfg["topk_groups"],
                       cfg["topk"], cfg["score_func"], cfg["route_scale"])
    weights, indices = routing
    nr = cfg["n_routed"]
    w13, s13, w2, s2 = p[pre + "w1w3_weight"], p[pre + "w1w3_scale"], p[pre + "w2_weight"], p[pre + "w2_scale"]

    def expert(e, xe):
        h = fp8.linear_deepseek_v3(xe, w13[e], s13[e])
        d = h.shape[-1] // 2
        return fp8.linear_deepseek_v3(F.silu(h[..., :d]) * h[..., d:], w2[e], s2[e])

    y = torch.zeros_like(x)
    for e in range(nr):
        idx, top = torch.where(indices == e)
        if idx.numel():
            y[idx] += expert(e, x[idx]) * weights[idx, top, None]
    for e in range(nr, w13.shape[0]):
        y += expert(e, x)
    return y, routing


def block(p, i, x, cos, sin, cache_layer, block_table, lens_excl, cfg, is_moe, routing=None):
    pre = f"layers.{i}."
    a, cache_layer = attention_decode(p, pre + "attn.", rms_norm(x, p[pre + "attn_norm.weight"], cfg["eps"]), cos, sin,
                                      cache_layer, block_table, lens_excl, cfg)
    x = x + a
    hn = rms_norm(x, p[pre + "ffn_norm.weight"], cfg["eps"])
    if is_moe:
        f, routing = (moe_layer_loop if cfg.get("moe_impl") == "loop" else moe_layer)(p, pre + "ffn.", hn, cfg, routing)
    else:
        f = mlp(p, pre + "ffn.", hn)
    return x + f, cache_layer, routing
