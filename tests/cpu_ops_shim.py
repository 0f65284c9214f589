"""TEST-ONLY: run chitu_amd.deepseek_v3's wiring on CPU by swapping every HIP op for the oracle.

The product has no CPU path (ops raise on CPU tensors).  To exercise the *host-side* logic of the
decode step -- module wiring, shapes, residual/pending handling, tensor-parallel sharding and the
placement of collectives -- without a GPU, the world-size-2 gloo tests monkeypatch the op entry
points with oracle-backed implementations.  Nothing outside tests/ imports this file.
"""

import torch
import torch.nn.functional as F

from oracle import deepseek as ods
from oracle import fp8 as ofp8
from oracle import kv as okv
from oracle import mla as omla
from oracle import moe as omoe


def rms_norm(x, weight, eps=1e-6, out_bf16=True, quant=None, add=None):
    res = ()
    if add is not None:
        if add.dim() == x.dim() + 1:  # un-summed top-k terms: moe_sum's arithmetic first
            add = add.float().sum(-2).to(x.dtype)
        x = x + add
        res = (x,)
    y = F.rms_norm(x, (x.shape[-1],), weight, eps).to(x.dtype)
    if quant is None:
        out = res + (y,)
        return out[0] if len(out) == 1 else out
    q, s = (ofp8.act_quant_deepseek_v3 if quant == "act" else ofp8.per_token_group_quant_fp8)(y.contiguous())
    return res + ((y if out_bf16 else None), q, s)


def act_quant_deepseek_v3(x, block_size=128):
    return ofp8.act_quant_deepseek_v3(x, block_size)


def fp8_gemm_deepseek_v3(a, a_s, b, b_s, out_dtype=None):
    return ofp8.fp8_gemm_deepseek_v3(a, a_s, b, b_s, out_dtype or torch.bfloat16)


def mla_kv_prep(kv_in, q_pe, cos, sin, kv_norm_weight, eps, kv_cache, page_table, old_seq_lens):
    qo, ko = okv.apply_rotary_pos_emb(q_pe, kv_in[:, 512:], cos, sin, "llama")
    q_pe.copy_(qo)
    kvn = F.rms_norm(kv_in[:, :512], (512,), kv_norm_weight, eps).to(kv_in.dtype)
    new = okv.append_to_paged_kv_cache(kv_cache, page_table, torch.cat([kvn, ko], -1), old_seq_lens)
    kv_cache.copy_(new)


def mla_qkv_post(q_a_kv, q_lora_rank, q_norm_weight, q_eps, kv_norm_weight, kv_eps, cos, sin, kv_cache, page_table,
                 old_seq_lens):
    _, qq, qs = rms_norm(q_a_kv[:, :q_lora_rank], q_norm_weight, q_eps, out_bf16=False, quant="act")
    kv_in = q_a_kv[:, q_lora_rank:]
    _, ko = okv.apply_rotary_pos_emb(kv_in[:, None, 512:].expand(-1, 1, -1), kv_in[:, 512:], cos, sin, "llama")
    kvn = F.rms_norm(kv_in[:, :512], (512,), kv_norm_weight, kv_eps).to(kv_in.dtype)
    kv_cache.copy_(okv.append_to_paged_kv_cache(kv_cache, page_table, torch.cat([kvn, ko], -1), old_seq_lens))
    return qq, qs


def mla_q_proj_fits(bs, q_lora_rank):
    return 0 < bs <= 32 and q_lora_rank % 128 == 0 and 128 <= q_lora_rank <= 2048


def mla_q_proj(q_a_kv, q_lora_rank, q_norm_weight, q_eps, wq_b, wq_b_scale, kv_norm_weight, kv_eps, cos, sin, kv_cache,
               page_table, old_seq_lens, out_dtype=None):
    qq, qs = mla_qkv_post(q_a_kv, q_lora_rank, q_norm_weight, q_eps, kv_norm_weight, kv_eps, cos, sin, kv_cache, page_table,
                          old_seq_lens)
    return fp8_gemm_deepseek_v3(qq, qs, wq_b, wq_b_scale, out_dtype=out_dtype)


def absorb_bmm_rope_fp8(x, w, scale, scale_offset, sh, sn, sk, q_pe, cos, sin):
    qo, _ = okv.apply_rotary_pos_emb(q_pe, q_pe[:, 0], cos, sin, "llama")
    q_pe.copy_(qo)
    return absorb_bmm_fp8(x, w, scale, scale_offset, sh, sn, sk)


def absorb_bmm_rope_kv_fp8(x, w, scale, scale_offset, sh, sn, sk, q_pe, cos, sin, kv_in, kv_norm_weight, eps, kv_cache,
                           page_table, old_seq_lens):
    mla_kv_prep(kv_in, q_pe, cos, sin, kv_norm_weight, eps, kv_cache, page_table, old_seq_lens)
    return absorb_bmm_fp8(x, w, scale, scale_offset, sh, sn, sk)


def _dequant_heads(w, scale, off, sh, sn, sk):
    H, N, K = w.shape
    out = torch.empty(H, N, K, dtype=torch.bfloat16)
    flat = scale.reshape(-1)
    for h in range(H):
        for nb in range((N + 127) // 128):
            for kb in range((K + 127) // 128):
                s = flat[off + h * sh + nb * sn + kb * sk]
                blk = w[h, nb * 128 : (nb + 1) * 128, kb * 128 : (kb + 1) * 128].float() * s
                out[h, nb * 128 : (nb + 1) * 128, kb * 128 : (kb + 1) * 128] = blk.to(torch.bfloat16)
    return out


def absorb_bmm_fp8(x, w, scale, scale_offset, sh, sn, sk):
    wd = _dequant_heads(w, scale, scale_offset, sh, sn, sk)
    return torch.einsum("bhk,hnk->bhn", x.float(), wd.float()).to(torch.bfloat16)


def absorb_uv_quant_fp8(x, w, scale, scale_offset, sh, sk):
    y = absorb_bmm_fp8(x, w, scale, scale_offset, sh, 0, sk).reshape(x.shape[0], -1)
    return ofp8.act_quant_deepseek_v3(y.contiguous())


def gate_deepseek_v3(x, weight, bias, n_groups, topk_groups, topk, score_func, route_scale, extra_expert_id=-1,
                     extra_weight=1.0, extra_count=1, align=None, logits_partials=None):
    assert logits_partials is None  # (the score GEMM always runs here)
    # align (the fused route + sort launch) is a launch-count optimisation: the 2-tuple makes the caller sort itself
    if score_func == "softmax_renorm":  # Mixtral: softmax -> top-k -> renormalise (model_hf_mixtral.py:60-75)
        from oracle import mixtral as omix

        return omix.route(x, weight, topk)
    w, i = ods.gate(x, weight, bias, n_groups, topk_groups, topk, score_func, route_scale)
    if extra_expert_id >= 0:
        w = torch.cat([w, torch.full((w.shape[0], extra_count), extra_weight, dtype=w.dtype)], 1)
        i = torch.cat([i, (extra_expert_id + torch.arange(extra_count, dtype=i.dtype)).expand(i.shape[0], -1)], 1)
    return w, i


def bf16_linear(x, weight, out_dtype=None):
    return F.linear(x, weight).to(out_dtype or torch.bfloat16)


def fused_experts(hidden_states, w1, w2, topk_weights, topk_ids, inplace=False, use_fp8_w8a8=False,
                  global_num_experts=-1, w1_scale=None, w2_scale=None, block_shape=None, a1_quant=None,
                  expert_map=None, use_int8_w8a8=False, **kw):
    if use_int8_w8a8:
        from oracle import w8a8 as ow

        out = ow.fused_experts_int8(hidden_states, w1, w2, topk_weights, topk_ids, w1_scale, w2_scale)
    else:
        out = omoe.fused_experts_fp8(hidden_states, w1, w2, topk_weights, topk_ids, w1_scale, w2_scale,
                                     expert_map=expert_map)
    if inplace:
        hidden_states.copy_(out)
        return hidden_states
    return out


def silu_and_mul_quant(x, mode="act"):
    d = x.shape[-1] // 2
    h = F.silu(x[..., :d]) * x[..., d:]
    return (ofp8.act_quant_deepseek_v3 if mode == "act" else ofp8.per_token_group_quant_fp8)(h.contiguous())


class CpuAttnBackend:
    def __init__(self, local_n_heads):
        self.local_n_heads = local_n_heads

    def prepare_metadata_for_decode(self, *a, **k):
        pass

    def mla_decode(self, q_nope, q_pe, kv_cache, lens_incl, block_table, softmax_scale, **kw):
        return omla.mla_decode(q_nope, q_pe, kv_cache, block_table, lens_incl, softmax_scale).to(torch.bfloat16)


# ---------------------------------------------------------------- Llama-family ops (chitu_amd/llama.py wiring on CPU)
def gqa_qkv_post(qkv, q_heads, kv_heads, cos, sin, k_cache, v_cache, page_table, old_seq_lens, rotary_type="llama"):
    from oracle import kv as okv_

    q, k = okv_.apply_rotary_pos_emb(qkv[:, :q_heads], qkv[:, q_heads : q_heads + kv_heads], cos, sin, rotary_type)
    v = qkv[:, q_heads + kv_heads :]
    page = k_cache.shape[1]
    for b in range(qkv.shape[0]):
        L = int(old_seq_lens[b])
        blk = int(page_table[b][L // page])
        k_cache[blk][L % page] = k[b]
        v_cache[blk][L % page] = v[b]
    return q.contiguous()


def bf16_linear_silu(x, w13):
    h = F.linear(x, w13)
    d = h.shape[-1] // 2
    return F.silu(h[..., :d]) * h[..., d:]


def bf16_add_norm_fits(M, N, K):
    return 1 <= M <= 4 and K % 64 == 0 and 512 <= K <= 8192 and M * K <= 24576


def bf16_linear_add_norm(x, add, norm_weight, eps, weight, out_dtype=None):
    x_new, y = rms_norm(x, norm_weight, eps, add=add)
    return x_new, bf16_linear(y, weight, out_dtype)


def bf16_linear_add_norm_qkv_post(x, add, norm_weight, eps, wqkv, q_heads, kv_heads, cos, sin, k_cache, v_cache,
                                  page_table, old_seq_lens):
    x_new, y = rms_norm(x, norm_weight, eps, add=add)
    d = wqkv.shape[0] // (q_heads + 2 * kv_heads)
    qkv = bf16_linear(y, wqkv).view(x.shape[0], q_heads + 2 * kv_heads, d)
    qkv = qkv.clone()
    qkv[:, :q_heads] = gqa_qkv_post(qkv, q_heads, kv_heads, cos, sin, k_cache, v_cache, page_table, old_seq_lens, rotary_type="llama")
    return x_new, qkv


def bf16_linear_silu_add_norm(x, add, norm_weight, eps, w13):
    x_new, y = rms_norm(x, norm_weight, eps, add=add)
    return x_new, bf16_linear_silu(y, w13)


def apply_rotary_pos_emb(q, k, cos, sin, rotary_type="hf-llama"):
    return okv.apply_rotary_pos_emb(q, k, cos, sin, rotary_type)


class CpuGqaBackend:
    """attn_with_kvcache / attn_varlen_func of HipAttnBackend on the oracle (oracle/gqa.py)."""

    def attn_with_kvcache(self, q, k_cache, v_cache, k=None, v=None, cache_seqlens=None, block_table=None, **kw):
        from oracle import gqa as ogqa

        out, kc, vc = ogqa.attn_with_kvcache(q, k_cache, v_cache, k, v, cache_seqlens, block_table)
        if k is not None:
            k_cache.copy_(kc)
            v_cache.copy_(vc)
        return out.to(torch.bfloat16)

    def attn_varlen_func(self, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, causal=False, **kw):
        from oracle import gqa as ogqa

        assert causal
        return ogqa.attn_varlen_causal(q, k, v, cu_seqlens_q).to(torch.bfloat16)


def embed_rope_gather(tokens, embed_weight, vocab_start, positions=None, cos_table=None, sin_table=None):
    """tensor_parallel.py:199-208 (lookup with foreign ids zeroed, no all-reduce) + model.py:429-448 (rotary rows)."""
    local = tokens - vocab_start
    foreign = (local < 0) | (local >= embed_weight.shape[0])
    h = torch.nn.functional.embedding(local.masked_fill(foreign, 0), embed_weight).masked_fill(foreign.unsqueeze(-1), 0)
    if positions is None:
        return h, None, None
    pos = positions[: tokens.shape[0]].long()
    return h, cos_table[pos], sin_table[pos]


def install_llama(monkeypatch_setattr):
    from chitu_amd import fused_moe, ops

    for name in ("rms_norm", "bf16_linear", "gqa_qkv_post", "bf16_linear_silu", "apply_rotary_pos_emb", "gate_deepseek_v3",
                 "embed_rope_gather", "bf16_add_norm_fits", "bf16_linear_add_norm", "bf16_linear_silu_add_norm",
                 "bf16_linear_add_norm_qkv_post"):
        monkeypatch_setattr(ops, name, globals()[name])
    monkeypatch_setattr(fused_moe, "fused_experts", fused_experts)  # Mixtral's INT8 experts ride on the Llama wiring


def fp8_linear_add_norm(x, add, norm_weight, eps, weight, weight_scale, out_dtype=torch.bfloat16):
    """ops.fp8_linear_add_norm as the composition it fuses ([terms sum +] add + norm + act_quant, then the W8A8 GEMM)."""
    x_new, _, q, s = rms_norm(x, norm_weight, eps, out_bf16=False, quant="act", add=add)
    return x_new, fp8_gemm_deepseek_v3(q, s, weight, weight_scale, out_dtype=out_dtype)


def tile_major_ok(rows):
    """The CPU shim keeps the reference's row-major (q, s) pairs: tile-major is a device-side layout."""
    return False


def install(monkeypatch_setattr):
    """monkeypatch_setattr(obj, name, value) -- e.g. pytest's monkeypatch.setattr or plain setattr."""
    from chitu_amd import fused_moe, ops

    for name in ("tile_major_ok", "rms_norm", "act_quant_deepseek_v3", "fp8_gemm_deepseek_v3", "mla_kv_prep", "absorb_bmm_fp8",
                 "absorb_uv_quant_fp8", "gate_deepseek_v3", "bf16_linear", "mla_qkv_post", "mla_q_proj", "mla_q_proj_fits", "absorb_bmm_rope_fp8",
                 "absorb_bmm_rope_kv_fp8", "fp8_linear_add_norm",
                 "embed_rope_gather"):
        monkeypatch_setattr(ops, name, globals()[name])
    monkeypatch_setattr(fused_moe, "fused_experts", fused_experts)
    monkeypatch_setattr(fused_moe, "silu_and_mul_quant", silu_and_mul_quant)
