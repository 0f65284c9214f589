"""CPU restatement of the tensor-parallel collectives of the decode step.  TEST INFRASTRUCTURE ONLY: nothing
under chitu_amd/ may import this.

Reference semantics: `dist.all_reduce(y)` = elementwise sum over the ranks (chitu/tensor_parallel.py:166,
chitu/models/model_deepseek_v3.py:1011), `all_gather_into_tensor` on the [N/tp, ...] layout = rank-major concat of
the last dimension (tensor_parallel.py:94-102).  NCCL fixes neither the summation order nor the intermediate
precision of a bf16 all-reduce, so the reference has no bit-level answer beyond world size 2 (where
bf16(a + b) is the only possibility).  The specification restated here -- and matched bit for bit by
csrc/comm.hip -- is the most accurate member of that family: fp32 accumulation in rank order, ONE rounding.
Parity status: pinned for world 2 (order-free); for world > 2 "within one bf16 rounding of any summation order".
"""

import torch

from . import deepseek as ods
from . import fp8 as ofp8


def moe_sum(part3: torch.Tensor) -> torch.Tensor:
    """fused_moe.py:1299-1305 (ops.moe_sum): fp32 sum over the top-k axis, one rounding."""
    return part3.float().sum(dim=1).to(torch.bfloat16)


def all_reduce(parts):
    """parts: list (rank order) of bf16 [rows, dim] or [rows, terms, dim] partials -> bf16 [rows, dim]."""
    acc = None
    for p in parts:
        p = moe_sum(p) if p.dim() == 3 else p
        acc = p.float() if acc is None else acc + p.float()
    return acc.to(torch.bfloat16)


def allreduce_rmsnorm(parts, x=None, weight=None, eps=1e-6, quant=None):
    """[top-k sum ->] all-reduce -> residual add (bf16, model_deepseek_v3.py:1107-1113) -> RMSNorm
    (models/model.py:29-78) -> fp8 quant of the rounded output (model_deepseek_v3.py:98-100 "act" rule,
    fused_moe.py:713-793 "group" rule).  Returns (x_new, y, q, s)."""
    a = all_reduce(parts)
    v = a if x is None else (x.float() + a.float()).to(torch.bfloat16)
    if weight is None:
        return v, None, None, None
    y = ods.rms_norm(v, weight, eps)
    q = s = None
    if quant == "act":
        q, s = ofp8.act_quant_deepseek_v3(y)
    elif quant == "group":
        q, s = ofp8.per_token_group_quant_fp8(y)
    return v, y, q, s


def all_gather_last_dim(parts, out_dtype=torch.bfloat16):
    return torch.cat(list(parts), dim=-1).to(out_dtype)
