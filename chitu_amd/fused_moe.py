"""Fused MoE for gfx950 with the reference's `chitu/fused_moe.py` surface.

Reference (read-only): chitu/fused_moe.py -- moe_align_block_size:599-610 (+ _cuda:445-519,
_native:522-596), per_token_group_quant_fp8:713-793, fused_experts:1060-1127,
fused_experts_impl:1130-1307, SiluAndMul:24-39.  No Triton here: every step is a hand-written
HIP kernel behind libchitu_hip.so, and there is no fallback path.
"""

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
    check(
        _lib.lib().chitu_hip_moe_align_block_size(
            ptr(topk_ids), int_dtype_code(topk_ids.dtype), i64(numel), i32(num_experts), i32(block_size),
            ptr(sorted_ids), i64(sorted_ids.numel()), ptr(expert_ids), i64(expert_ids.numel()),
            ptr(num_tokens_post_pad), ptr(cumsum_buffer), i32(1), stream_ptr(),
        ),
        "moe_align_block_size",
    )
    if expert_map is not None:
        expert_ids = expert_map[expert_ids]
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
