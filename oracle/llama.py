"""Oracle for one Llama decode layer (torch CPU).  TEST INFRASTRUCTURE ONLY.

Composition of pinned pieces, in the order of chitu/models/model.py (Attention.decode_forward_paged
:167-198, FeedForward.forward :212-214, TransformerBlock.forward): RMSNorm = F.rms_norm (:29-78),
linears = F.linear on bf16 (fp32 accumulate, one rounding), RoPE = oracle/kv.py (pinned against the
reference kernel), attention = oracle/gqa.py (pinned against RefAttnBackend), SiluAndMul on bf16.
"""

import torch
import torch.nn.functional as F

from . import gqa as ogqa
from . import kv as okv


def rms_norm(x, w, eps):
    return F.rms_norm(x, (x.shape[-1],), w, eps).to(x.dtype)


def block(p, pre, x, cos, sin, k_cache, v_cache, block_table, lens_excl, hq, hkv, hd, eps, rotary="llama", ffn=None):
    """x [bs, dim] bf16 -> (x_out, k_cache', v_cache').  ffn: optional replacement of the dense SwiGLU MLP
    (hn -> f), e.g. the Mixtral sparse MoE of oracle/mixtral.py."""
    bs = x.shape[0]
    hn = rms_norm(x, p[pre + "attn_norm"], eps)
    qkv = F.linear(hn, p[pre + "attn.wqkv"]).view(bs, hq + 2 * hkv, hd)
    q, k = okv.apply_rotary_pos_emb(qkv[:, :hq], qkv[:, hq : hq + hkv], cos, sin, rotary)
    v = qkv[:, hq + hkv :]
    o, k_cache, v_cache = ogqa.attn_with_kvcache(q.reshape(bs, 1, hq, hd), k_cache, v_cache, k.reshape(bs, 1, hkv, hd),
                                                 v.reshape(bs, 1, hkv, hd), lens_excl, block_table)
    a = F.linear(o.to(torch.bfloat16).view(bs, hq * hd), p[pre + "attn.wo"])
    x = x + a
    hn = rms_norm(x, p[pre + "ffn_norm"], eps)
    if ffn is not None:
        return x + ffn(hn), k_cache, v_cache
    h13 = F.linear(hn, p[pre + "ffn.w13"])
    d = h13.shape[-1] // 2
    f = F.linear(F.silu(h13[..., :d]) * h13[..., d:], p[pre + "ffn.w2"])
    return x + f, k_cache, v_cache


def decode_sequence(p, tokens, n_layers, hq, hkv, hd, eps, theta, page=256, rotary="llama"):
    """A whole (tiny) Llama on ONE sequence, token by token through `block` over a paged cache: embedding, layers,
    final norm, head -> fp32 logits [len(tokens), vocab] (models/model.py:468-475 decode_single_device per step).
    Pinned against the reference's own TransformerLlama run (BASELINE config 1: tests/golden/ref_llama.npz); the
    reference prefills the prompt in one varlen pass, which differs from this token-by-token form only in fp32
    summation order."""
    n = len(tokens)
    nblk = (n + page - 1) // page
    k_cache = [torch.zeros(nblk, page, hkv, hd, dtype=torch.bfloat16) for _ in range(n_layers)]
    v_cache = [torch.zeros(nblk, page, hkv, hd, dtype=torch.bfloat16) for _ in range(n_layers)]
    table = torch.arange(nblk, dtype=torch.int32).view(1, nblk)
    freqs = 1.0 / (theta ** (torch.arange(0, hd, 2)[: hd // 2].float() / hd))  # models/model.py:81-88
    cis = torch.polar(torch.ones(n, hd // 2), torch.outer(torch.arange(n, dtype=torch.float32), freqs))
    out = []
    for pos, tok in enumerate(tokens):
        x = p["embed_weight"][int(tok)].view(1, -1)
        cos, sin = cis.real[pos : pos + 1].contiguous(), cis.imag[pos : pos + 1].contiguous()
        lens = torch.tensor([pos], dtype=torch.int32)
        for i in range(n_layers):
            x, k_cache[i], v_cache[i] = block(p, f"layers.{i}.", x, cos, sin, k_cache[i], v_cache[i], table, lens,
                                              hq, hkv, hd, eps, rotary)
        out.append(F.linear(rms_norm(x, p["norm"], eps), p["head_weight"]).float())
    return torch.cat(out)
