"""Oracle for MLA absorb-mode paged decode attention (torch CPU fp32).  TEST INFRASTRUCTURE ONLY.

Restates chitu/triton_decode_attention.py:20-130 (+ :185-232 merge) as plain softmax attention
over the gathered pages, the way RefAttnBackend._attention does it for the non-paged cache
(chitu/attn_backend.py:294-392): scores = scale * (q_nope . c + q_pe . k_pe), out = softmax . c.
The split-KV / LSE merge of the reference is mathematically the identity on this result.
"""

import torch


def gather_pages(cache, block_table_row, length):
    """cache [P, page, D] -> [length, D] following the block table (attn_backend.py:716-760)."""
    page = cache.shape[1]
    rows = []
    for t in range(0, length, page):
        n = min(page, length - t)
        rows.append(cache[int(block_table_row[t // page])][:n])
    return torch.cat(rows, dim=0) if rows else cache.new_zeros(0, cache.shape[-1])


def mla_decode(q_nope, q_pe, cache, block_table, seqlens, scale, kv_lora_rank=512):
    """q_nope [bs,H,C], q_pe [bs,H,R], cache [P,page,C+R], seqlens incl. this token -> [bs,H,C] fp32."""
    bs, H, C = q_nope.shape
    out = torch.zeros(bs, H, C, dtype=torch.float32)
    for b in range(bs):
        L = int(seqlens[b])
        if L == 0:
            continue
        kv = gather_pages(cache, block_table[b], L).float()
        c, pe = kv[:, :kv_lora_rank], kv[:, kv_lora_rank:]
        s = (q_nope[b].float() @ c.T + q_pe[b].float() @ pe.T) * scale
        p = torch.softmax(s, dim=-1)
        out[b] = p @ c
    return out


def mla_attn_with_kvcache(q_nope, q_pe, kv_cache, kv, seqlens_excl, seqlens_incl, block_table, scale):
    """attn_backend.py:707-774: append this token's row to its page, then decode."""
    from . import kv as okv

    cache = okv.append_to_paged_kv_cache(kv_cache, block_table, kv, seqlens_excl)
    return mla_decode(q_nope, q_pe, cache, block_table, seqlens_incl, scale), cache


def mla_prefill(q, kv, cu_seqlens, scale, kv_lora_rank=512):
    """Causal MQA over each sequence's own keys, values = the latent part of the keys -- the call
    AttentionDeepSeekV3.prefill_forward makes in absorb mode (model_deepseek_v3.py:589-599:
    attn_varlen_func(q_nope|q_pe, kv|k_pe, kv, prefix_lens, ...)), with RefAttnBackend._attention's
    math (attn_backend.py:294-392).  q [T, H, C+R], kv [T, C+R] -> [T, H, C] fp32.
    Pinned by tests/golden/mla_prefill.npz (RefAttnBackend.attn_varlen_func run in the build container)."""
    T, H, _ = q.shape
    out = torch.zeros(T, H, kv_lora_rank, dtype=torch.float32)
    cu = [int(c) for c in cu_seqlens]
    for s0, s1 in zip(cu[:-1], cu[1:]):
        k = kv[s0:s1].float()
        n = s1 - s0
        scores = torch.einsum("thd,sd->hts", q[s0:s1].float() * scale, k)
        mask = torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1)
        scores.masked_fill_(mask, float("-inf"))
        out[s0:s1] = torch.einsum("hts,sc->thc", torch.softmax(scores, dim=-1), k[:, :kv_lora_rank])
    return out
