"""Oracle for the Mixtral sparse-MoE block with INT8 W8A8 experts (torch CPU).  TEST INFRASTRUCTURE ONLY.

Router verbatim from SparseMoeBlockHFMixtral.forward (chitu/models/model_hf_mixtral.py:53-64): bf16
logits, softmax in fp32, top-k, renormalise, cast to the activation dtype; experts = the per-expert
W8A8Linear loop of oracle/w8a8.py::fused_experts_int8 (what `simple_w8a8` makes of the reference's expert
loop).  No reference fixture for the fused form (SURVEY gap G2); the quantisers are pinned bit-exactly.
"""

import torch
import torch.nn.functional as F

from . import w8a8 as ow


def route(x, gate_w, topk):
    logits = F.linear(x, gate_w)
    w = torch.softmax(logits, dim=-1, dtype=torch.float)
    w, ids = torch.topk(w, topk, dim=-1)
    w = w / w.sum(dim=-1, keepdim=True)
    return w.to(x.dtype), ids


def sparse_moe(p, pre, x, topk, routing=None):
    w, ids = route(x, p[pre + "gate"], topk) if routing is None else routing
    return ow.fused_experts_int8(x, p[pre + "w13"], p[pre + "w2"], w, ids, p[pre + "w13_scale"], p[pre + "w2_scale"]), (w, ids)
