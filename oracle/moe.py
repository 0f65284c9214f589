"""Oracle for the fused MoE (fp8 block-scaled W8A8 path), torch CPU.  TEST INFRASTRUCTURE ONLY.

Restates chitu/fused_moe.py:1130-1307 (fused_experts_impl) as a per-(token, slot) loop with the
reference's rounding points: activations re-quantised to e4m3 per 128-group before each GEMM
(:829), GEMM outputs rounded to the activation dtype (:299-301), SiLU-and-mul in that dtype
(:37-39), routed weight applied on the fp32 accumulator of GEMM2 (:290-292), top-k sum (:1299).
"""

import torch
import torch.nn.functional as F

from . import fp8


def fused_experts_fp8(x, w1, w2, topk_weights, topk_ids, w1_scale, w2_scale, expert_map=None):
    """x [M,K] bf16; w1 [E,2I,K] fp8; w2 [E,K,I] fp8; scales [E, rows/128, cols/128] f32.
    expert_map [global experts] (expert parallelism, fused_moe.py:449,516-517): topk_ids are GLOBAL ids,
    expert_map[id] is the row of w1/w2 on this rank or -1; a slot whose expert is elsewhere contributes
    zeros (write_zeros_to_output, fused_moe.py:40-59, 163-179)."""
    if expert_map is not None:
        local = torch.as_tensor(expert_map).long()[topk_ids.long()]
        keep = local >= 0
        out = fused_experts_fp8(x, w1, w2, torch.where(keep, topk_weights, torch.zeros_like(topk_weights)),
                                torch.where(keep, local, torch.zeros_like(local)), w1_scale, w2_scale)
        # a zero routed weight makes the slot's bf16 term exactly +-0 -- the same sum as skipping it
        return out
    M, K = x.shape
    topk = topk_ids.shape[1]
    dt = x.dtype
    a1_q, a1_s = fp8.per_token_group_quant_fp8(x)
    c1 = torch.empty(M, topk, w1.shape[1], dtype=dt)
    for t in range(M):
        for j in range(topk):
            e = int(topk_ids[t, j])
            c1[t, j] = fp8.fp8_gemm_deepseek_v3(a1_q[t : t + 1], a1_s[t : t + 1], w1[e], w1_scale[e], dt)[0]
    d = w1.shape[1] // 2
    c1 = c1.view(-1, w1.shape[1])
    c2 = F.silu(c1[..., :d]) * c1[..., d:]
    a2_q, a2_s = fp8.per_token_group_quant_fp8(c2)
    c3 = torch.empty(M, topk, w2.shape[1], dtype=dt)
    for t in range(M):
        for j in range(topk):
            e = int(topk_ids[t, j])
            r = t * topk + j
            acc = fp8.fp8_gemm_deepseek_v3(a2_q[r : r + 1], a2_s[r : r + 1], w2[e], w2_scale[e], torch.float32)[0]
            c3[t, j] = fp8.to_out(acc * topk_weights[t, j].float(), dt)
    return c3.sum(dim=1)


def fused_experts_bf16(x, w1, w2, topk_weights, topk_ids, expert_map=None):
    """The UNQUANTISED branch of the fused MoE (fused_experts_impl with use_fp8_w8a8=False: fused_moe.py:298
    `accumulator += tl.dot(a, b)`, compute_type = the activation dtype).  x [M,K], w1 [E,2I,K], w2 [E,N,I] in
    one float dtype (bf16 / f16 / f32).  Rounding points: GEMM outputs -> dtype (:299-306), SiluAndMul as torch ops
    on that dtype (:24-39, :1262-1265), routed weight on GEMM2's fp32 accumulator (:290-292), top-k sum as
    `input.sum(dim=1)` in that dtype (:1299-1305).  expert_map: as fused_experts_fp8."""
    if expert_map is not None:
        local = torch.as_tensor(expert_map).long()[topk_ids.long()]
        keep = local >= 0
        return fused_experts_bf16(x, w1, w2, torch.where(keep, topk_weights, torch.zeros_like(topk_weights)),
                                  torch.where(keep, local, torch.zeros_like(local)))
    M, K = x.shape
    topk = topk_ids.shape[1]
    dt = x.dtype
    c1 = torch.empty(M, topk, w1.shape[1], dtype=dt)
    for t in range(M):
        for j in range(topk):
            c1[t, j] = fp8.to_out(x[t].float() @ w1[int(topk_ids[t, j])].float().T, dt)
    d = w1.shape[1] // 2
    c1 = c1.view(-1, w1.shape[1])
    c2 = F.silu(c1[..., :d]) * c1[..., d:]
    c3 = torch.empty(M, topk, w2.shape[1], dtype=dt)
    for t in range(M):
        for j in range(topk):
            acc = c2[t * topk + j].float() @ w2[int(topk_ids[t, j])].float().T
            c3[t, j] = fp8.to_out(acc * topk_weights[t, j].float(), dt)
    return c3.sum(dim=1)


def fused_experts_soft_fp8(x, w1, w2, topk_weights, topk_ids, w1_scale, w2_scale, expert_map=None):
    """fused_experts_impl(use_fp8_w8a8=True, soft_fp8=True) (fused_moe.py:232-276): activations stay in their dtype,
    every fp8 weight is decoded to bf16( bits(w) * (scale * 2^120) ) -- weight_dequant_soft_fp8's arithmetic, element
    for element -- and multiplied as bf16.  Equal to the bf16 branch on the dequantised weights, which is literally
    what the reference does on non-NVIDIA devices (model_deepseek_v3.py:975-993).  PARITY NOTE: the fused soft kernel
    itself holds PTX and cannot run here; its two pinned pieces are the decode (tests/test_oracle_golden.py, soft
    dequant fixture) and the bf16 branch (fused_moe_bf16.npz)."""
    w1d = torch.stack([fp8.weight_dequant_soft_fp8_deepseek_v3(w1[e], w1_scale[e], torch.bfloat16) for e in range(w1.shape[0])])
    w2d = torch.stack([fp8.weight_dequant_soft_fp8_deepseek_v3(w2[e], w2_scale[e], torch.bfloat16) for e in range(w2.shape[0])])
    return fused_experts_bf16(x, w1d.to(x.dtype), w2d.to(x.dtype), topk_weights, topk_ids, expert_map)
