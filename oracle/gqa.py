"""Oracle for paged GQA/MHA single-token decode attention (torch CPU fp32).  TEST INFRASTRUCTURE ONLY.

The reference's PAGED path is the third-party flash_attn.flash_attn_with_kvcache (absent here,
`chitu/attn_backend.py:208-243`) and RefAttnBackend rejects block_table (:473), so there is no reference
run of the paged layout itself.  This restates the documented contract (attn_backend.py:92-164: in-place
append at cache_seqlens, GQA head mapping "head i of Q attends head i // g of KV") with the math of
RefAttnBackend._attention (:294-392) on the gathered pages, and is PINNED on the arithmetic by
tests/golden/gqa_decode.npz: RefAttnBackend.attn_with_kvcache (:457-516) run in the build container on
contiguous caches holding the same logical content the test lays out in (shuffled) pages.
"""

import torch


def gather(cache, table_row, length):
    page = cache.shape[1]
    rows = [cache[int(table_row[t // page])][: min(page, length - t)] for t in range(0, length, page)]
    return torch.cat(rows, 0) if rows else cache.new_zeros((0,) + tuple(cache.shape[2:]))


def attn_with_kvcache(q, k_cache, v_cache, k, v, cache_seqlens, block_table, softmax_scale=None):
    """q [bs,1,Hq,D]; caches [P,page,Hkv,D]; k/v [bs,1,Hkv,D] appended at cache_seqlens; returns (out, k_cache, v_cache)."""
    bs, _, Hq, D = q.shape
    Hkv = k_cache.shape[2]
    g = Hq // Hkv
    page = k_cache.shape[1]
    k_cache, v_cache = k_cache.clone(), v_cache.clone()
    scale = softmax_scale if softmax_scale is not None else D ** -0.5
    out = torch.zeros(bs, 1, Hq, D, dtype=torch.float32)
    for b in range(bs):
        L = int(cache_seqlens[b])
        if k is not None:
            k_cache[int(block_table[b][L // page])][L % page] = k[b, 0]
            v_cache[int(block_table[b][L // page])][L % page] = v[b, 0]
            L += 1
        if L == 0:
            continue
        K = gather(k_cache, block_table[b], L).float()  # [L, Hkv, D]
        V = gather(v_cache, block_table[b], L).float()
        for h in range(Hq):
            s = (K[:, h // g] @ q[b, 0, h].float()) * scale
            p = torch.softmax(s, dim=0)
            out[b, 0, h] = p @ V[:, h // g]
    return out, k_cache, v_cache


def attn_varlen_causal(q, k, v, cu_seqlens, softmax_scale=None):
    """Causal GQA self-attention per sequence (Attention.prefill_forward, models/model.py:104-132, with
    RefAttnBackend._attention's math): q [T, Hq, D], k / v [T, Hkv, D] -> [T, Hq, D] fp32.
    Pinned by tests/golden/gqa_prefill.npz (RefAttnBackend.attn_varlen_func run in the build container)."""
    T, Hq, D = q.shape
    g = Hq // k.shape[1]
    scale = softmax_scale if softmax_scale is not None else D ** -0.5
    out = torch.zeros(T, Hq, D, dtype=torch.float32)
    cu = [int(c) for c in cu_seqlens]
    for s0, s1 in zip(cu[:-1], cu[1:]):
        n = s1 - s0
        kk = k[s0:s1].float().repeat_interleave(g, dim=1)
        vv = v[s0:s1].float().repeat_interleave(g, dim=1)
        sc = torch.einsum("thd,shd->hts", q[s0:s1].float() * scale, kk)
        sc.masked_fill_(torch.triu(torch.ones(n, n, dtype=torch.bool), diagonal=1), float("-inf"))
        out[s0:s1] = torch.einsum("hts,shd->thd", torch.softmax(sc, dim=-1), vv)
    return out
