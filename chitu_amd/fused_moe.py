"""Fused MoE for gfx950 with the reference's `chitu/fused_moe.py` surface.

Reference (read-only): chitu/fused_moe.py -- moe_align_block_size:599-610 (+ _cuda:445-519,
_native:522-596), per_token_group_quant_fp8:713-793, fused_experts:1060-1127,
fused_experts_impl:1130-1307, SiluAndMul:24-39.  No Triton here: every step is a hand-written
HIP kernel behind libchitu_hip.so, and there is no fallback path.
"""

import os
from typing import List, Optional, Tuple

import torch

from . import _lib, workspace
from ._lib import check, f32, float_dtype_code, i32, i64, int_dtype_code, ptr, require_cuda, stream_ptr

__all__ = [
    "moe_align_block_size",
    "per_token_group_quant_fp8",
    "fused_experts",
    "fused_experts_impl",
]


def ceil_div(a, b):
    return (a + b - 1) // b


def cuda_moe_align_block_size(
    topk_ids: torch.Tensor,
    num_experts: int,
    block_size: int,
    sorted_token_ids: torch.Tensor,
    experts_ids: torch.Tensor,
    num_tokens_post_pad: torch.Tensor,
    cumsum_buffer: torch.Tensor,
) -> None:
    """Drop-in for `chitu_backend.cuda_moe_align_block_size` (csrc/binding.cpp:11, moe_kernel.h:7-11).

    Same 7 arguments, same caller-allocated/pre-filled buffers; unlike the CUDA kernel the
    order inside an expert segment is stable (== the reference's Triton path).
    """
    for t in (topk_ids, sorted_token_ids, experts_ids, num_tokens_post_pad, cumsum_buffer):
        if not t.is_contiguous():
            raise RuntimeError("Tensor is not contiguous")  # csrc/common.h:46-55
    require_cuda(topk_ids, sorted_token_ids, experts_ids, num_tokens_post_pad, cumsum_buffer)
    for t in (sorted_token_ids, experts_ids, num_tokens_post_pad, cumsum_buffer):
        if t.dtype != torch.int32:
            raise RuntimeError("Tensor type is incorrect")
    if cumsum_buffer.numel() < num_experts + 1:
        raise RuntimeError("cumsum_buffer too small")
    check(
        _lib.lib().chitu_hip_moe_align_block_size(
            ptr(topk_ids), int_dtype_code(topk_ids.dtype), i64(topk_ids.numel()), i32(num_experts),
            i32(block_size), ptr(sorted_token_ids), i64(sorted_token_ids.numel()), ptr(experts_ids),
            i64(experts_ids.numel()), ptr(num_tokens_post_pad), ptr(cumsum_buffer), i32(0), stream_ptr(),
        ),
        "cuda_moe_align_block_size",
    )


def _expert_map_i32(expert_map: Optional[torch.Tensor], num_experts: int, dev) -> Optional[torch.Tensor]:
    """expert_map [global_num_experts] -> contiguous int32 on the device (local id, or -1 for an expert that
    lives on another rank: fused_moe.py:163-179).  Pass an int32 device tensor to make this a no-op."""
    if expert_map is None:
        return None
    assert expert_map.numel() >= num_experts, "expert_map must cover every global expert id"
    if expert_map.dtype != torch.int32 or expert_map.device != dev or not expert_map.is_contiguous():
        expert_map = expert_map.to(device=dev, dtype=torch.int32).contiguous()
    return expert_map


def moe_align_block_size(
    topk_ids: torch.Tensor,
    block_size: int,
    num_experts: int,
    expert_map: torch.Tensor = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Stable counting sort of the flat top-k ids by expert, segments padded to block_size.

    Returns (sorted_ids, expert_ids, num_tokens_post_pad) with the reference's allocation
    contract (fused_moe.py:489-519): sorted_ids has numel + E*(block-1) entries, padding slots
    hold `numel`; expert_ids has one entry per block, unused entries 0.  One launch: the
    kernel writes the sentinels itself.
    """
    require_cuda(topk_ids)
    if not topk_ids.is_contiguous():
        topk_ids = topk_ids.contiguous()
    numel = topk_ids.numel()
    max_num_tokens_padded = numel + num_experts * (block_size - 1)
    dev = topk_ids.device
    sorted_ids = torch.empty((max_num_tokens_padded,), dtype=torch.int32, device=dev)
    max_num_m_blocks = ceil_div(max_num_tokens_padded, block_size)
    expert_ids = torch.empty((max_num_m_blocks,), dtype=torch.int32, device=dev)
    num_tokens_post_pad = torch.empty((1), dtype=torch.int32, device=dev)
    cumsum_buffer = torch.empty((num_experts + 1,), dtype=torch.int32, device=dev)
    emap = _expert_map_i32(expert_map, num_experts, dev)
    check(
        _lib.lib().chitu_hip_moe_align_block_size_mapped(
            ptr(topk_ids), int_dtype_code(topk_ids.dtype), i64(numel), i32(num_experts), i32(block_size),
            ptr(sorted_ids), i64(sorted_ids.numel()), ptr(expert_ids), i64(expert_ids.numel()),
            ptr(num_tokens_post_pad), ptr(cumsum_buffer), i32(1), ptr(emap), stream_ptr(),
        ),
        "moe_align_block_size",
    )  # expert_ids = expert_map[expert_ids] (fused_moe.py:516-517) happens inside the launch
    return sorted_ids, expert_ids, num_tokens_post_pad


def per_token_group_quant_fp8(
    x: torch.Tensor,
    group_size: int,
    eps: float = 1e-10,
    dtype: Optional[torch.dtype] = None,
    column_major_scales: bool = False,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """s = max(max|x|, eps)/448 per group, q = clamp(x/s, +-448) -> e4m3fn (fused_moe.py:713-793)."""
    if dtype is None:
        dtype = torch.float8_e4m3fn
    assert dtype == torch.float8_e4m3fn, "only torch.float8_e4m3fn (OCP) is supported"
    assert x.shape[-1] % group_size == 0, (
        f"the last dimension of `x` {x.shape[-1]} must be divisible " f"by `group_size` {group_size}"
    )
    assert x.stride(-1) == 1, "`x` groups must be contiguous"
    assert not column_major_scales, "column-major scales are not used on this path"
    require_cuda(x)
    if not x.is_contiguous():
        x = x.contiguous()
    x_q = torch.empty_like(x, dtype=dtype)
    x_s = torch.empty(x.shape[:-1] + (x.shape[-1] // group_size,), device=x.device, dtype=torch.float32)
    cols = x.shape[-1]
    rows = x.numel() // cols if cols else 0
    check(
        _lib.lib().chitu_hip_act_quant_fp8(
            ptr(x), float_dtype_code(x.dtype), i64(rows), i64(cols), i32(group_size), i32(1), f32(eps),
            ptr(x_q), ptr(x_s), stream_ptr(),
        ),
        "per_token_group_quant_fp8",
    )
    return x_q, x_s


_MOE_BLOCK_M = 16  # one MFMA tile of sorted slots (the reference's BLOCK_SIZE_M=64 is >90% padding in decode)
# Prefill: once the experts see dozens of tokens each, the grouped GEMMs are tiled for compute -- moe_align with the
# reference's block 64, 64-slot x 128-row tiles through LDS (csrc/moe_tiled.hip).  Taken from _MOE_TILED_MIN_TOKENS tokens
# on (0 = never) when the average expert holds at least _MOE_TILED_MIN_PER_EXPERT slots (below that the 64-slot tiles are
# mostly padding: R1 rank shard, 128 prompt tokens = 4 slots per expert: 0.50 -> 0.56 ms per layer; 512 tokens = 16:
# neutral; 2048 tokens = 64: 2.03 -> 1.70 ms).
_MOE_TILED_MIN_TOKENS = int(os.environ.get("CHITU_MOE_TILED_MIN_TOKENS", "128"))
_MOE_TILED_MIN_PER_EXPERT = 24
# Slots per tile = moe_align block of the tiled path: 128 (round 6: one pass over an expert's weights for up to 128 slots, padding
# sub-tiles skipped) or 64 (rounds 2-5, the reference's BLOCK_SIZE_M); same outputs, A/B in profiles/r06_ab_moe_tiled128.txt
_MOE_TILED_BLOCK_M = int(os.environ.get("CHITU_MOE_TILED_BLOCK_M", "128"))


class SiluAndMul(torch.nn.Module):
    """x -> silu(x[..., :d]) * x[..., d:]  (chitu/fused_moe.py:24-39).  Kept for API parity; the
    fused path applies it inside chitu_hip_moe_silu_mul_quant_fp8."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        d = x.shape[-1] // 2
        return torch.nn.functional.silu(x[..., :d]) * x[..., d:]


def fused_experts(
    hidden_states: torch.Tensor,
    w1: torch.Tensor,
    w2: torch.Tensor,
    topk_weights: torch.Tensor,
    topk_ids: torch.Tensor,
    inplace: bool = False,
    activation: str = "silu",
    use_fp8_w8a8: bool = False,
    use_int8_w8a16: bool = False,
    use_int4_w4a16: bool = False,
    global_num_experts: int = -1,
    expert_map: Optional[torch.Tensor] = None,
    w1_scale: Optional[torch.Tensor] = None,
    w2_scale: Optional[torch.Tensor] = None,
    w1_zp: Optional[torch.Tensor] = None,
    w2_zp: Optional[torch.Tensor] = None,
    a1_scale: Optional[torch.Tensor] = None,
    a2_scale: Optional[torch.Tensor] = None,
    block_shape: Optional[List[int]] = None,
    soft_fp8: bool = False,
    a1_quant=None,
    reduce_topk: bool = True,
    use_int8_w8a8: bool = False,
    aligned=None,
) -> torch.Tensor:
    """Same signature as chitu/fused_moe.py:1060-1127 (+ optional a1_quant / reduce_topk, see fused_experts_impl).  (The reference's inplace=False branch
    calls an unregistered torch.ops.vllm op; here both branches work.)"""
    return fused_experts_impl(
        hidden_states, w1, w2, topk_weights, topk_ids, inplace, activation, use_fp8_w8a8,
        use_int8_w8a16, use_int4_w4a16, global_num_experts, expert_map, w1_scale, w2_scale, w1_zp,
        w2_zp, a1_scale, a2_scale, block_shape, soft_fp8=soft_fp8, a1_quant=a1_quant, reduce_topk=reduce_topk,
        use_int8_w8a8=use_int8_w8a8, aligned=aligned,
    )


def fused_experts_impl(
    hidden_states: torch.Tensor,
    w1: torch.Tensor,
    w2: torch.Tensor,
    topk_weights: torch.Tensor,
    topk_ids: torch.Tensor,
    inplace: bool = False,
    activation: str = "silu",
    use_fp8_w8a8: bool = False,
    use_int8_w8a16: bool = False,
    use_int4_w4a16: bool = False,
    global_num_experts: int = -1,
    expert_map: Optional[torch.Tensor] = None,
    w1_scale: Optional[torch.Tensor] = None,
    w2_scale: Optional[torch.Tensor] = None,
    w1_zp: Optional[torch.Tensor] = None,
    w2_zp: Optional[torch.Tensor] = None,
    a1_scale: Optional[torch.Tensor] = None,
    a2_scale: Optional[torch.Tensor] = None,
    block_shape: Optional[List[int]] = None,
    soft_fp8: bool = False,
    a1_quant=None,
    reduce_topk: bool = True,
    use_int8_w8a8: bool = False,
    aligned=None,
):
    """out[t] = sum_j w[t,j] * W2[e_tj] . (silu(W1[e_tj] x_t)[:I] * (W1[e_tj] x_t)[I:])

    Modes (fused_moe.py:1130-1307, the branches of fused_moe_kernel:216-298): FP8 W8A8 with [128,128] block scales
    (use_fp8_w8a8=True: the DeepSeek-V3/R1 path, described below), the same weights with bf16 activations
    (soft_fp8=True) and bf16 experts (use_fp8_w8a8=False) -- `_fused_experts_bf16_act` --, INT8 W8A8 (use_int8_w8a8).
    Launches: align(16) -> [quant ->] grouped GEMM1 (+ silu*mul) -> grouped GEMM2 (requant +, x routed
    weight) -> top-k sum.  Scratch lives in a persistent workspace (graph-capture safe).
    a1_quant=(q, s): the per-128-group fp8 form of hidden_states if the producer (fused RMSNorm)
    already computed it -- skips the quant launch, numerics unchanged.
    reduce_topk=False: skip the top-k sum and return the routed-weighted expert outputs
    [tokens, topk, hidden] (a view of the persistent workspace, valid until the next fused_experts
    call) for a consumer that sums them itself (ops.rms_norm(add=<3-D>), same arithmetic).
    aligned=(sorted_token_ids, expert_ids, num_tokens_post_pad): moe_align_block_size(topk_ids, 16,
    global_num_experts, expert_map) already computed by the producer of topk_ids
    (ops.gate_deepseek_v3(align=...): routing and sort in one launch) -- skips the align launch.
    """
    assert hidden_states.shape[1] == w1.shape[2], "Hidden size mismatch"
    assert topk_weights.shape == topk_ids.shape, "topk shape mismatch"
    assert hidden_states.is_contiguous(), "Hidden_states must be contiguous"
    assert w1.is_contiguous(), "Expert weights1 must be contiguous"
    assert w2.is_contiguous(), "Expert weights2 must be contiguous"
    assert hidden_states.dtype in [torch.float32, torch.float16, torch.bfloat16]
    if activation != "silu":
        raise ValueError(f"Unsupported FusedMoe activation: {activation}")
    if use_int8_w8a8:
        return _fused_experts_int8(hidden_states, w1, w2, topk_weights, topk_ids, inplace, global_num_experts,
                                   expert_map, w1_scale, w2_scale, reduce_topk, aligned, a1_quant)
    if use_int8_w8a16 or use_int4_w4a16:
        raise NotImplementedError(
            "chitu_amd.fused_moe implements the modes the reference's DeepSeek MoE drives (bf16 experts, fp8_w8a8 "
            "block-scaled experts with soft_fp8 on or off) and int8 W8A8 (use_int8_w8a8=True); the weight-only "
            "int8 / int4 modes are not built"
        )
    if not use_fp8_w8a8 or soft_fp8:
        return _fused_experts_bf16_act(hidden_states, w1, w2, topk_weights, topk_ids, inplace, global_num_experts,
                                       expert_map, w1_scale if use_fp8_w8a8 else None,
                                       w2_scale if use_fp8_w8a8 else None, block_shape, reduce_topk, aligned)
    assert block_shape is not None and list(block_shape) == [128, 128], "block_shape must be [128, 128]"
    assert w1_scale is not None and w2_scale is not None
    assert a1_scale is None and a2_scale is None, "dynamic per-token-group activation scales only"
    assert hidden_states.dtype == torch.bfloat16, "bf16 activations (the reference's R1 configuration)"
    require_cuda(hidden_states, w1, w2, topk_weights, topk_ids, w1_scale, w2_scale)
    assert w1_scale.is_contiguous() and w2_scale.is_contiguous()
    assert w1_scale.dtype == torch.float32 and w2_scale.dtype == torch.float32

    num_tokens, K = hidden_states.shape
    E, N, _ = w1.shape
    I = N // 2
    assert w2.shape[0] == E and w2.shape[2] == I, "w2 must be [E, hidden_out, N/2]"
    Nout = w2.shape[1]
    if global_num_experts == -1:
        global_num_experts = E
    topk = topk_ids.shape[1]
    dev = hidden_states.device
    out = hidden_states if inplace else torch.empty_like(hidden_states)
    if num_tokens == 0:
        return out
    numel = num_tokens * topk
    if not topk_ids.is_contiguous():
        topk_ids = topk_ids.contiguous()
    if not topk_weights.is_contiguous():
        topk_weights = topk_weights.contiguous()

    # ---- scratch carve-up (persistent, 256-B aligned)
    def rnd(n):
        return (n + 255) // 256 * 256

    tiled = (_MOE_TILED_MIN_TOKENS > 0 and num_tokens >= _MOE_TILED_MIN_TOKENS and aligned is None and I % 128 == 0
             and Nout % 8 == 0 and numel >= _MOE_TILED_MIN_PER_EXPERT * global_num_experts)
    block_m = _MOE_TILED_BLOCK_M if tiled else _MOE_BLOCK_M
    cap = numel + global_num_experts * (block_m - 1)
    nblk = ceil_div(cap, block_m)
    KB = K // 128
    sizes = [
        ("sorted", cap * 4), ("experts", nblk * 4), ("npost", 4), ("cumsum", (global_num_experts + 1) * 4),
        ("a1q", num_tokens * K), ("a1s", num_tokens * KB * 4), ("c1", numel * N * 2),
        ("a2q", numel * I), ("a2s", numel * (I // 128) * 4), ("c3", numel * Nout * 2),
    ]
    total = sum(rnd(n) for _, n in sizes)
    ws = workspace.get(total, dev, "moe")
    base = ws.data_ptr()
    off = {}
    cur = 0
    for name, n in sizes:
        off[name] = base + cur
        cur += rnd(n)
    import ctypes as _ct

    P = lambda name: _ct.c_void_p(off[name])
    lib = _lib.lib()
    st = stream_ptr()
    max_mblocks = min(nblk, numel)

    if aligned is None:
        emap = _expert_map_i32(expert_map, global_num_experts, dev)
        check(
            lib.chitu_hip_moe_align_block_size_mapped(
                ptr(topk_ids), int_dtype_code(topk_ids.dtype), i64(numel), i32(global_num_experts),
                i32(block_m), P("sorted"), i64(cap), P("experts"), i64(nblk), P("npost"), P("cumsum"),
                i32(1), ptr(emap), st,
            ),
            "moe_align_block_size",
        )
        sorted_p, experts_ptr, npost_p = P("sorted"), P("experts"), P("npost")
    else:
        a_sorted, a_experts, a_npost = aligned
        require_cuda(a_sorted, a_experts, a_npost)
        assert a_sorted.dtype == torch.int32 and a_experts.dtype == torch.int32 and a_npost.dtype == torch.int32
        assert a_sorted.numel() == cap and a_experts.numel() == nblk, "aligned buffers must come from block 16 over global_num_experts"
        sorted_p, experts_ptr, npost_p = ptr(a_sorted), ptr(a_experts), ptr(a_npost)
    if a1_quant is None:
        check(
            lib.chitu_hip_act_quant_fp8(
                ptr(hidden_states), float_dtype_code(hidden_states.dtype), i64(num_tokens), i64(K), i32(128),
                i32(1), f32(1e-10), P("a1q"), P("a1s"), st,
            ),
            "moe quant1",
        )
        a1q_p, a1s_p = P("a1q"), P("a1s")
    else:
        aq, as_ = a1_quant
        assert aq.is_contiguous() and as_.is_contiguous() and aq.numel() == num_tokens * K
        assert as_.dtype == torch.float32 and as_.numel() == num_tokens * KB
        a1q_p, a1s_p = ptr(aq), ptr(as_)
    if tiled:
        # prefill: GEMM1 + SiLU-and-mul tiled, per-token-group quant of h, GEMM2 tiled (csrc/moe_tiled.hip)
        assert nblk <= 65535
        check(
            lib.chitu_hip_moe_gemm1_silu_fp8_tiled(
                a1q_p, a1s_p, ptr(w1), ptr(w1_scale), sorted_p, experts_ptr, npost_p, P("c1"),
                i64(numel), i32(topk), i64(I), i64(K), i64(max_mblocks), i32(block_m), st,
            ),
            "moe gemm1 (tiled, silu fused)",
        )
        check(
            lib.chitu_hip_act_quant_fp8(P("c1"), float_dtype_code(torch.bfloat16), i64(numel), i64(I), i32(128), i32(1),
                                        f32(1e-10), P("a2q"), P("a2s"), st),
            "moe quant2",
        )
        check(
            lib.chitu_hip_moe_gemm2_fp8_tiled(
                P("a2q"), P("a2s"), ptr(w2), ptr(w2_scale), sorted_p, experts_ptr, npost_p,
                ptr(topk_weights), float_dtype_code(topk_weights.dtype), i32(1), P("c3"), i64(numel), i64(Nout),
                i64(I), i64(max_mblocks), i32(block_m), st,
            ),
            "moe gemm2 (tiled)",
        )
    elif I % 128 == 0 and I <= 512 and os.environ.get("CHITU_MOE_FUSE_SILU", "1") != "0":
        # two launches: GEMM1 with SiLU-and-mul in its epilogue (gate and up tile of the same columns
        # in one wave), GEMM2 with the fp8 re-quantisation of h in its prologue.  Experts wider than 512 take the three-launch
        # form: every workgroup of an m-block repeats the quantisation of its 16 x I activations before its first MFMA, which a
        # 512-wide expert hides and a 1408-wide one does not (profiles/r03_v2lite_wide_experts.txt).
        check(
            lib.chitu_hip_moe_gemm1_silu_fp8(
                a1q_p, a1s_p, ptr(w1), ptr(w1_scale), sorted_p, experts_ptr, npost_p, P("c1"),
                i64(numel), i32(topk), i64(I), i64(K), i64(max_mblocks), st,
            ),
            "moe gemm1 (silu fused)",
        )
        check(
            lib.chitu_hip_moe_gemm2_quant_fp8(
                P("c1"), ptr(w2), ptr(w2_scale), sorted_p, experts_ptr, npost_p, ptr(topk_weights),
                float_dtype_code(topk_weights.dtype), i32(1), P("c3"), i64(numel), i64(Nout), i64(I),
                i64(max_mblocks), f32(1e-10), st,
            ),
            "moe gemm2 (quant fused)",
        )
    else:
        # wide experts: GEMM1, SiLU + quant, GEMM2.  (GEMM1 with SiLU-and-mul AND the fp8 re-quantisation in its epilogue was
        # built in round 3 and measured slower at V2-Lite's shapes -- 64.5 us against 50.7 + 4.95, profiles/r03_v2lite_wide_experts.txt:
        # its 8-wave workgroups halve the resident waves per CU -- and removed in round 5.)
        check(
            lib.chitu_hip_moe_gemm1_fp8(
                a1q_p, a1s_p, ptr(w1), ptr(w1_scale), sorted_p, experts_ptr, npost_p, P("c1"),
                i64(numel), i32(topk), i64(N), i64(K), i64(max_mblocks), st,
            ),
            "moe gemm1",
        )
        check(
            lib.chitu_hip_moe_silu_mul_quant_fp8(P("c1"), i64(numel), i64(I), i32(1), f32(1e-10), P("a2q"), P("a2s"), st),
            "moe silu_mul_quant",
        )
        check(
            lib.chitu_hip_moe_gemm2_fp8(
                P("a2q"), P("a2s"), ptr(w2), ptr(w2_scale), sorted_p, experts_ptr, npost_p,
                ptr(topk_weights), float_dtype_code(topk_weights.dtype), i32(1), P("c3"), i64(numel), i64(Nout),
                i64(I), i64(max_mblocks), st,
            ),
            "moe gemm2",
        )
    if not reduce_topk:
        c3_off = off["c3"] - base
        return ws[c3_off : c3_off + numel * Nout * 2].view(torch.bfloat16).view(num_tokens, topk, Nout)
    check(lib.chitu_hip_moe_sum(P("c3"), ptr(out), i64(num_tokens), i32(topk), i64(Nout), st), "moe sum")
    return out


def _fused_experts_int8(hidden_states, w1, w2, topk_weights, topk_ids, inplace, global_num_experts, expert_map,
                        w1_scale, w2_scale, reduce_topk, aligned=None, a1_quant=None):
    """INT8 W8A8 experts (Mixtral + simple_w8a8, BASELINE config 4): per-token int8 activations, per-channel
    int8 weights.  w1 [E, 2I, K] int8, w1_scale [E, 2I]; w2 [E, N, I] int8, w2_scale [E, N].
    align(16) -> quant_act -> grouped GEMM1 (+ silu*mul) -> quant_act -> grouped GEMM2 (x routed weight) -> sum:
    the per-expert W8A8Linear arithmetic of the reference's expert loop (model_hf_mixtral.py:76-94,
    quantize/w8a8.py:97-132), grouped."""
    assert hidden_states.dtype == torch.bfloat16 and w1.dtype == torch.int8 and w2.dtype == torch.int8
    assert w1_scale is not None and w2_scale is not None and w1_scale.dtype == torch.float32 and w2_scale.dtype == torch.float32
    require_cuda(hidden_states, w1, w2, topk_weights, topk_ids, w1_scale, w2_scale)
    num_tokens, K = hidden_states.shape
    E, N, _ = w1.shape
    I = N // 2
    Nout = w2.shape[1]
    assert w2.shape[0] == E and w2.shape[2] == I and tuple(w1_scale.shape) == (E, N) and tuple(w2_scale.shape) == (E, Nout)
    assert w1_scale.is_contiguous() and w2_scale.is_contiguous() and K % 128 == 0 and I % 128 == 0
    if global_num_experts == -1:
        global_num_experts = E
    topk = topk_ids.shape[1]
    dev = hidden_states.device
    out = hidden_states if inplace else torch.empty_like(hidden_states)
    if num_tokens == 0:
        return out
    numel = num_tokens * topk
    topk_ids = topk_ids.contiguous()
    topk_weights = topk_weights.contiguous()
    cap = numel + global_num_experts * (_MOE_BLOCK_M - 1)
    nblk = ceil_div(cap, _MOE_BLOCK_M)
    rnd = lambda n: (n + 255) // 256 * 256
    sizes = [("sorted", cap * 4), ("experts", nblk * 4), ("npost", 4), ("cumsum", (global_num_experts + 1) * 4),
             ("xq", num_tokens * K), ("xs", num_tokens * 4), ("a", numel * I * 2), ("aq", numel * I), ("as", numel * 4),
             ("c3", numel * Nout * 2)]
    ws = workspace.get(sum(rnd(n) for _, n in sizes), dev, "moe")
    base, off, cur = ws.data_ptr(), {}, 0
    for name, n in sizes:
        off[name] = base + cur
        cur += rnd(n)
    import ctypes as _ct

    P = lambda name: _ct.c_void_p(off[name])
    lib, st = _lib.lib(), stream_ptr()
    max_mblocks = min(nblk, numel)
    if aligned is None:
        emap = _expert_map_i32(expert_map, global_num_experts, hidden_states.device)
        check(lib.chitu_hip_moe_align_block_size_mapped(ptr(topk_ids), int_dtype_code(topk_ids.dtype), i64(numel),
                                                        i32(global_num_experts), i32(_MOE_BLOCK_M), P("sorted"), i64(cap),
                                                        P("experts"), i64(nblk), P("npost"), P("cumsum"), i32(1), ptr(emap), st),
              "moe_align_block_size")
        sorted_p, experts_p, npost_p = P("sorted"), P("experts"), P("npost")
    else:  # the router's launch already sorted the ids (ops.gate_deepseek_v3(align=...)): one launch less per layer
        a_sorted, a_experts, a_npost = aligned
        require_cuda(a_sorted, a_experts, a_npost)
        assert a_sorted.dtype == torch.int32 and a_experts.dtype == torch.int32 and a_npost.dtype == torch.int32
        assert a_sorted.numel() == cap and a_experts.numel() == nblk, "aligned buffers must come from block 16 over global_num_experts"
        sorted_p, experts_p, npost_p = ptr(a_sorted), ptr(a_experts), ptr(a_npost)
    if a1_quant is None:
        check(lib.chitu_hip_quant_act_int8(ptr(hidden_states), float_dtype_code(hidden_states.dtype), i64(num_tokens), i64(K),
                                           P("xq"), P("xs"), st), "moe int8 quant1")
        xq_p, xs_p = P("xq"), P("xs")
    else:  # the producer (ops.rms_norm(quant="int8")) already quantised the tokens: same codes, one launch less
        aq_, as_ = a1_quant
        require_cuda(aq_, as_)
        assert aq_.dtype == torch.int8 and aq_.is_contiguous() and aq_.numel() == num_tokens * K
        assert as_.dtype == torch.float32 and as_.is_contiguous() and as_.numel() == num_tokens
        xq_p, xs_p = ptr(aq_), ptr(as_)
    check(lib.chitu_hip_moe_i8_gemm1_silu(xq_p, xs_p, ptr(w1), ptr(w1_scale), sorted_p, experts_p, npost_p, P("a"),
                                          i64(numel), i32(topk), i64(I), i64(K), i64(max_mblocks), st), "moe int8 gemm1")
    check(lib.chitu_hip_quant_act_int8(P("a"), i32(0), i64(numel), i64(I), P("aq"), P("as"), st), "moe int8 quant2")
    check(lib.chitu_hip_moe_i8_gemm2(P("aq"), P("as"), ptr(w2), ptr(w2_scale), sorted_p, experts_p, npost_p,
                                     ptr(topk_weights), float_dtype_code(topk_weights.dtype), i32(1), P("c3"), i64(numel),
                                     i64(Nout), i64(I), i64(max_mblocks), st), "moe int8 gemm2")
    if not reduce_topk:
        c3_off = off["c3"] - base
        return ws[c3_off : c3_off + numel * Nout * 2].view(torch.bfloat16).view(num_tokens, topk, Nout)
    check(lib.chitu_hip_moe_sum(P("c3"), ptr(out), i64(num_tokens), i32(topk), i64(Nout), st), "moe sum")
    return out


def _fused_experts_bf16_act(hidden_states, w1, w2, topk_weights, topk_ids, inplace, global_num_experts, expert_map,
                            w1_scale, w2_scale, block_shape, reduce_topk, aligned):
    """bf16 activations: bf16 experts (w*_scale None; use_fp8_w8a8=False, fused_moe.py:298) or fp8 experts decoded to
    bf16 in registers (soft_fp8=True, fused_moe.py:232-276).  align(16) -> grouped GEMM1 with SiluAndMul in its epilogue
    -> grouped GEMM2 (x routed weight) -> top-k sum: the reference's four steps (fused_moe.py:1230-1305) with its
    rounding points (bf16 after each GEMM, SiluAndMul on bf16 tensors), no activation quantisation."""
    soft = w1_scale is not None
    assert hidden_states.dtype == torch.bfloat16, "bf16 activations"
    if soft:
        assert block_shape is not None and list(block_shape) == [128, 128], "block_shape must be [128, 128]"
        assert w2_scale is not None and w1.element_size() == 1 and w2.element_size() == 1
        assert w1_scale.dtype == torch.float32 and w2_scale.dtype == torch.float32
        assert w1_scale.is_contiguous() and w2_scale.is_contiguous()
    else:
        assert w1.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16, "bf16 expert weights"
    require_cuda(hidden_states, w1, w2, topk_weights, topk_ids, w1_scale, w2_scale)
    num_tokens, K = hidden_states.shape
    E, N, _ = w1.shape
    I = N // 2
    assert w2.shape[0] == E and w2.shape[2] == I, "w2 must be [E, hidden_out, N/2]"
    Nout = w2.shape[1]
    assert K % 128 == 0 and I % 128 == 0, "K and the expert width must be multiples of 128"
    if global_num_experts == -1:
        global_num_experts = E
    topk = topk_ids.shape[1]
    dev = hidden_states.device
    out = hidden_states if inplace else torch.empty_like(hidden_states)
    if num_tokens == 0:
        return out
    numel = num_tokens * topk
    topk_ids = topk_ids.contiguous()
    topk_weights = topk_weights.contiguous()
    cap = numel + global_num_experts * (_MOE_BLOCK_M - 1)
    nblk = ceil_div(cap, _MOE_BLOCK_M)
    rnd = lambda n: (n + 255) // 256 * 256
    sizes = [("sorted", cap * 4), ("experts", nblk * 4), ("npost", 4), ("cumsum", (global_num_experts + 1) * 4),
             ("h", numel * I * 2), ("c3", numel * Nout * 2)]
    ws = workspace.get(sum(rnd(n) for _, n in sizes), dev, "moe")
    base, off, cur = ws.data_ptr(), {}, 0
    for name, n in sizes:
        off[name] = base + cur
        cur += rnd(n)
    import ctypes as _ct

    P = lambda name: _ct.c_void_p(off[name])
    lib, st = _lib.lib(), stream_ptr()
    max_mblocks = min(nblk, numel)
    if aligned is None:
        emap = _expert_map_i32(expert_map, global_num_experts, dev)
        check(lib.chitu_hip_moe_align_block_size_mapped(ptr(topk_ids), int_dtype_code(topk_ids.dtype), i64(numel),
                                                        i32(global_num_experts), i32(_MOE_BLOCK_M), P("sorted"), i64(cap),
                                                        P("experts"), i64(nblk), P("npost"), P("cumsum"), i32(1), ptr(emap), st),
              "moe_align_block_size")
        sorted_p, experts_ptr, npost_p = P("sorted"), P("experts"), P("npost")
    else:
        a_sorted, a_experts, a_npost = aligned
        require_cuda(a_sorted, a_experts, a_npost)
        assert a_sorted.numel() == cap and a_experts.numel() == nblk, "aligned buffers must come from block 16 over global_num_experts"
        sorted_p, experts_ptr, npost_p = ptr(a_sorted), ptr(a_experts), ptr(a_npost)
    kind = i32(1 if soft else 0)
    check(lib.chitu_hip_moe_gemm_bf16(ptr(hidden_states), i32(topk), ptr(w1), ptr(w1_scale), kind, sorted_p, experts_ptr,
                                      npost_p, ptr(None), i32(0), i32(0), i32(1), P("h"), i64(numel), i64(I), i64(K),
                                      i64(max_mblocks), st), "moe gemm1 (bf16 activations, silu fused)")
    check(lib.chitu_hip_moe_gemm_bf16(P("h"), i32(1), ptr(w2), ptr(w2_scale), kind, sorted_p, experts_ptr, npost_p,
                                      ptr(topk_weights), float_dtype_code(topk_weights.dtype), i32(1), i32(0), P("c3"),
                                      i64(numel), i64(Nout), i64(I), i64(max_mblocks), st), "moe gemm2 (bf16 activations)")
    if not reduce_topk:
        c3_off = off["c3"] - base
        return ws[c3_off : c3_off + numel * Nout * 2].view(torch.bfloat16).view(num_tokens, topk, Nout)
    check(lib.chitu_hip_moe_sum(P("c3"), ptr(out), i64(num_tokens), i32(topk), i64(Nout), st), "moe sum")
    return out


def silu_and_mul_quant(x: torch.Tensor, mode: str = "act"):
    """h = silu(x[..., :d]) * x[..., d:] (SiluAndMul, fused_moe.py:24-39) followed by the 128-group
    fp8 quantisation the next fp8 GEMM needs.  mode "act": act_quant_deepseek_v3 rule (dense and
    shared-expert MLPs, model_deepseek_v3.py:771, 936-949); "group": per_token_group_quant_fp8."""
    require_cuda(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous()
    d = x.shape[-1] // 2
    rows = x.numel() // x.shape[-1]
    q = torch.empty(*x.shape[:-1], d, dtype=torch.float8_e4m3fn, device=x.device)
    s = torch.empty(*x.shape[:-1], d // 128, dtype=torch.float32, device=x.device)
    check(
        _lib.lib().chitu_hip_moe_silu_mul_quant_fp8(
            ptr(x), i64(rows), i64(d), i32(0 if mode == "act" else 1), f32(1e-10), ptr(q), ptr(s), stream_ptr()
        ),
        "silu_and_mul_quant",
    )
    return q, s
