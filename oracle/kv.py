"""Oracle for paged-KV append and RoPE (torch CPU).  TEST INFRASTRUCTURE ONLY."""

import torch

from . import fp8 as _fp8  # CAST["out"]: the final bf16 rounding (swappable: the Triton interpreter truncates)


def append_to_paged_kv_cache(kv_cache, page_table, this_kv, old_seq_lens):
    """chitu/ops.py:57-60 docstring / triton_kernels.py:18-48 (page arithmetic uses the page size;
    the reference kernel hard-codes 64, which is the MLA page size)."""
    page = kv_cache.shape[1]
    out = kv_cache.clone()
    flat = this_kv.reshape(this_kv.shape[0], -1)
    for i in range(old_seq_lens.shape[0]):
        L = int(old_seq_lens[i])
        out[int(page_table[i][L // page])][L % page] = flat[i].reshape(kv_cache.shape[2:])
    return out


def apply_rotary_pos_emb(q, k, cos, sin, rotary_type="llama"):
    """chitu/ops.py:243-272 (apply_rotary_pos_emb_torch) restated without broadcasting helpers.

    q: [bs, heads, d]; k: [bs, d] or [bs, kv_heads, d]; cos/sin: [bs, d/2] fp32.
    "llama" = interleaved (re, im) pairs; "hf-llama" = half split.
    """

    def rot(x):
        xf = x.float()
        c = cos.view(cos.shape[0], *([1] * (x.dim() - 2)), cos.shape[-1])
        s = sin.view(sin.shape[0], *([1] * (x.dim() - 2)), sin.shape[-1])
        if rotary_type == "llama":
            x0, x1 = xf[..., 0::2], xf[..., 1::2]
            o0 = x0 * c + (-x1) * s
            o1 = x1 * c + x0 * s
            return _fp8.CAST["out"](torch.stack([o0, o1], dim=-1).flatten(-2), x.dtype)
        elif rotary_type == "hf-llama":
            h = xf.shape[-1] // 2
            x0, x1 = xf[..., :h], xf[..., h:]
            o0 = x0 * c + (-x1) * s
            o1 = x1 * c + x0 * s
            return _fp8.CAST["out"](torch.cat([o0, o1], dim=-1), x.dtype)
        raise ValueError(rotary_type)

    return rot(q), rot(k)
