"""HIP launchers with the reference's `chitu/ops.py` names, signatures and asserts.

Reference (read-only): chitu/ops.py:51-511.  Each function below calls one C-ABI entry
point of libchitu_hip.so on torch's current stream.  There is no Triton and no fallback.
"""

import os as _os
from typing import Tuple

import torch

from . import _lib, workspace
from ._lib import check, f32, float_dtype_code, i32, i64, ptr, require_cuda, stream_ptr

__all__ = [
    "append_to_paged_kv_cache",
    "apply_rotary_pos_emb",
    "apply_rotary_pos_emb_torch",
    "act_quant_deepseek_v3",
    "weight_dequant_deepseek_v3",
    "weight_dequant_soft_fp8_deepseek_v3",
    "fp8_gemm_deepseek_v3",
    "soft_fp8_gemm_deepseek_v3",
]

_GEMM_WS_BYTES = 64 << 20
_TILE_MAJOR_MAX_ROWS = 64
_TILE_MAJOR = _os.environ.get("CHITU_TILE_MAJOR", "1") != "0"  # read once: A/B timing and the equivalence test flip it


class TiledQuant:
    """fp8 activations in the TILE-MAJOR layout of the fused decode step: q bytes [ceil(rows/16)][K/16][16 rows][16 B],
    scales [ceil(rows/16)][K/128][16 rows] -- what a small-batch GEMM reads with fully coalesced loads (a 16-lane group =
    256 contiguous bytes; row-major it is 16 B from each of 16 rows).  Written by rms_norm(add=, quant=, tile_major=True)
    and mla_merge_absorb_uv_quant_fp8(tile_major=True), read by fp8_gemm_deepseek_v3(TiledQuant, ...).  Internal to the
    fused step: the drop-in op surface keeps the reference's row-major [rows, K] / [rows, K/128] pair."""

    __slots__ = ("q", "s", "rows", "cols")

    def __init__(self, q, s, rows, cols):
        self.q, self.s, self.rows, self.cols = q, s, rows, cols

    def to_row_major(self):
        """(q [rows, K] fp8, s [rows, K/128] f32): the same codes and scales in the reference's layout (tests)."""
        t = (self.rows + 15) // 16
        q = self.q.view(torch.uint8).view(t, self.cols // 16, 16, 16).permute(0, 2, 1, 3).reshape(t * 16, self.cols)
        s = self.s.view(t, self.cols // 128, 16).permute(0, 2, 1).reshape(t * 16, self.cols // 128)
        return q[: self.rows].contiguous().view(torch.float8_e4m3fn), s[: self.rows].contiguous()


def tile_major_ok(rows: int) -> bool:
    """Do the fused step's quantising launches write their fp8 output tile-major for a batch of `rows`?  (Decode-sized
    batches only: from _TILE_MAJOR_MAX_ROWS on the GEMMs are tiled for compute and read row-major.)"""
    return _TILE_MAJOR and 0 < rows <= _TILE_MAJOR_MAX_ROWS


def _tiled_buffers(rows, cols, device):
    t = (rows + 15) // 16
    return (torch.empty(t * 16, cols, dtype=torch.float8_e4m3fn, device=device),
            torch.empty(t, cols // 128, 16, dtype=torch.float32, device=device))


def act_quant_deepseek_v3(x: torch.Tensor, block_size: int = 128) -> Tuple[torch.Tensor, torch.Tensor]:
    """Block-wise e4m3 quantisation: s = max|x|/448 per 128 contiguous values, y = x/s.

    Same contract as chitu/ops.py:330-353 (kernel triton_kernels.py:193-214): no eps and no
    clamp, so an all-zero group yields scale 0 and NaN codes exactly like the reference.
    """
    assert x.is_contiguous(), "Input tensor must be contiguous"
    assert (
        x.size(-1) % block_size == 0
    ), f"Last dimension size must be divisible by block_size (block_size={block_size})"
    require_cuda(x)
    y = torch.empty_like(x, dtype=torch.float8_e4m3fn)
    s = x.new_empty(*x.size()[:-1], x.size(-1) // block_size, dtype=torch.float32)
    cols = x.size(-1)
    rows = x.numel() // cols if cols else 0
    check(
        _lib.lib().chitu_hip_act_quant_fp8(
            ptr(x), float_dtype_code(x.dtype), i64(rows), i64(cols), i32(block_size), i32(0),
            f32(0.0), ptr(y), ptr(s), stream_ptr(),
        ),
        "act_quant_deepseek_v3",
    )
    return y, s


def _weight_dequant(x, s, block_size, soft):
    assert x.is_contiguous() and s.is_contiguous(), "Input tensors must be contiguous"
    assert s.dim() == x.dim(), "Scale tensors must have the same number of dimensions with the weight tensor"
    if x.dim() == 2:
        M, N = x.size()
        B = 1
    elif x.dim() == 3:
        B, M, N = x.size()
    else:
        assert False, "Weight tensor must have 2 or 3 dimensions"
    require_cuda(x, s)
    assert x.element_size() == 1 and s.dtype == torch.float32, "fp8 weight + float32 block scales expected"
    y = torch.empty_like(x, dtype=torch.get_default_dtype())
    check(
        _lib.lib().chitu_hip_weight_dequant_fp8(
            ptr(x), ptr(s), i64(B), i64(M), i64(N), i32(block_size), i32(1 if soft else 0),
            float_dtype_code(y.dtype), ptr(y), stream_ptr(),
        ),
        "weight_dequant",
    )
    return y


def weight_dequant_deepseek_v3(x: torch.Tensor, s: torch.Tensor, block_size: int = 128) -> torch.Tensor:
    """y = float(x) * s[block] in the default dtype (chitu/ops.py:357-392)."""
    return _weight_dequant(x, s, block_size, soft=False)


def weight_dequant_soft_fp8_deepseek_v3(x: torch.Tensor, s: torch.Tensor, block_size: int = 128) -> torch.Tensor:
    """The reference's bit-placement decode (chitu/ops.py:396-449), one fused kernel here."""
    return _weight_dequant(x, s, block_size, soft=True)


def fp8_gemm_deepseek_v3(a: torch.Tensor, a_s: torch.Tensor, b: torch.Tensor, b_s: torch.Tensor, out_dtype=None):
    """c = sum_kb (a_kb . b_kb^T) * a_s[:, kb] * b_s[n//128, kb]   (chitu/ops.py:453-483).
    Output dtype = torch.get_default_dtype() like the reference (:474), unless out_dtype is given.
    a may be a TiledQuant (then a_s is ignored): the same GEMM reading tile-major activations, bit-identical."""
    if isinstance(a, TiledQuant):
        assert b.is_contiguous() and b_s.is_contiguous() and b.element_size() == 1 and b_s.dtype == torch.float32
        require_cuda(a.q, a.s, b, b_s)
        assert a.cols == b.size(-1)
        c = torch.empty(a.rows, b.size(0), dtype=out_dtype or torch.get_default_dtype(), device=b.device)
        ws = workspace.get(_GEMM_WS_BYTES, b.device, "gemm")
        check(
            _lib.lib().chitu_hip_fp8_gemm_blockscale_tm(
                ptr(a.q), ptr(a.s), ptr(b), ptr(b_s), ptr(c), float_dtype_code(c.dtype), i64(a.rows), i64(b.size(0)),
                i64(a.cols), ptr(ws), i64(ws.numel()), stream_ptr(),
            ),
            "fp8_gemm_deepseek_v3 (tile-major activations)",
        )
        return c
    assert a.is_contiguous() and b.is_contiguous(), "Input tensors must be contiguous"
    assert a_s.is_contiguous() and b_s.is_contiguous(), "Scaling factor tensors must be contiguous"
    require_cuda(a, a_s, b, b_s)
    assert a.element_size() == 1 and b.element_size() == 1, "fp8 operands expected"
    assert a_s.dtype == torch.float32 and b_s.dtype == torch.float32, "float32 scales expected"
    K = a.size(-1)
    M = a.numel() // K
    N = b.size(0)
    c = a.new_empty(*a.size()[:-1], N, dtype=out_dtype or torch.get_default_dtype())
    ws = workspace.get(_GEMM_WS_BYTES, a.device, "gemm")
    check(
        _lib.lib().chitu_hip_fp8_gemm_blockscale(
            ptr(a), ptr(a_s), ptr(b), ptr(b_s), ptr(c), float_dtype_code(c.dtype), i64(M), i64(N),
            i64(K), ptr(ws), i64(ws.numel()), stream_ptr(),
        ),
        "fp8_gemm_deepseek_v3",
    )
    return c


def fp8_gemm_partials_deepseek_v3(a: torch.Tensor, a_s: torch.Tensor, b: torch.Tensor, b_s: torch.Tensor, num_splits: int):
    """fp8_gemm_deepseek_v3 with the K range cut over `num_splits` workgroups per tile; returns the fp32 partial
    planes [num_splits, M, N] (sum over dim 0, rounded once, = the GEMM output) for a consumer that folds the
    sum into its own loads (mla_qkv_post(num_partials=...))."""
    assert a.is_contiguous() and b.is_contiguous() and a_s.is_contiguous() and b_s.is_contiguous()
    require_cuda(a, a_s, b, b_s)
    assert a.element_size() == 1 and b.element_size() == 1 and a_s.dtype == torch.float32 and b_s.dtype == torch.float32
    K = a.size(-1)
    M = a.numel() // K
    N = b.size(0)
    parts = torch.empty(num_splits, M, N, dtype=torch.float32, device=a.device)
    check(
        _lib.lib().chitu_hip_fp8_gemm_blockscale_partials(ptr(a), ptr(a_s), ptr(b), ptr(b_s), ptr(parts), i64(M), i64(N),
                                                          i64(K), i32(num_splits), stream_ptr()),
        "fp8_gemm_partials_deepseek_v3",
    )
    return parts


def soft_fp8_gemm_deepseek_v3(a: torch.Tensor, b: torch.Tensor, b_s: torch.Tensor):
    """FP8 weights decoded to bf16 on the fly, bf16 x bf16 dot (chitu/ops.py:487-511)."""
    assert a.is_contiguous() and b.is_contiguous(), "Input tensors must be contiguous"
    assert b_s.is_contiguous(), "Scaling factor tensor must be contiguous"
    assert a.dtype == torch.bfloat16, "soft-fp8 GEMM takes bf16 activations"
    require_cuda(a, b, b_s)
    assert b.element_size() == 1 and b_s.dtype == torch.float32
    K = a.size(-1)
    M = a.numel() // K
    N = b.size(0)
    c = a.new_empty(*a.size()[:-1], N, dtype=torch.get_default_dtype())
    ws = workspace.get(_GEMM_WS_BYTES, a.device, "gemm")
    check(
        _lib.lib().chitu_hip_soft_fp8_gemm(
            ptr(a), ptr(b.view(torch.uint8)), ptr(b_s), ptr(c), float_dtype_code(c.dtype), i64(M),
            i64(N), i64(K), ptr(ws), i64(ws.numel()), stream_ptr(),
        ),
        "soft_fp8_gemm_deepseek_v3",
    )
    return c


# ---- paged KV append and RoPE are defined further down (kv.hip) -------------------------


def append_to_paged_kv_cache(kv_cache, page_table, this_kv, old_seq_lens):
    """kv_cache[page_table[i][L_i // page]][L_i % page] = this_kv[i]   (chitu/ops.py:51-91).

    The reference kernel hard-codes 64 in the page arithmetic (triton_kernels.py:38,42); here
    the page size is kv_cache.shape[1], which is identical for the MLA cache (page 64).
    """
    assert kv_cache.is_contiguous()
    assert page_table.is_contiguous()
    assert this_kv.is_contiguous()
    assert old_seq_lens.is_contiguous()
    require_cuda(kv_cache, page_table, this_kv, old_seq_lens)
    page_size = kv_cache.shape[1]
    batch_size, num_pages_per_sample = page_table.shape
    assert this_kv.shape[0] == batch_size
    assert old_seq_lens.shape[0] == batch_size
    row = this_kv.numel() // batch_size if batch_size else 0
    assert kv_cache.numel() // (kv_cache.shape[0] * kv_cache.shape[1]) == row
    assert page_table.dtype == torch.int32 and old_seq_lens.dtype == torch.int32
    assert this_kv.dtype == kv_cache.dtype
    check(
        _lib.lib().chitu_hip_append_paged_kv(
            ptr(kv_cache), i64(kv_cache.shape[0]), i32(page_size), i64(row * kv_cache.element_size()),
            ptr(page_table), i32(num_pages_per_sample), ptr(this_kv), ptr(old_seq_lens),
            i32(batch_size), stream_ptr(),
        ),
        "append_to_paged_kv_cache",
    )


def apply_rotary_pos_emb_torch(q, k, cos, sin, rotary_type="hf-llama"):
    """Name kept for source compatibility with chitu/ops.py:243-308; runs the HIP kernel."""
    return apply_rotary_pos_emb(q, k, cos, sin, rotary_type=rotary_type)


def apply_rotary_pos_emb(q, k, cos, sin, rotary_type="hf-llama"):
    """RoPE on q [bs, heads, d] and k [bs, d] or [bs, kv_heads, d]   (chitu/ops.py:311-326).

    rotary_type "llama": interleaved (re, im) pairs (DeepSeek MLA); "hf-llama": half-split.
    cos/sin are fp32 [bs, d/2].  Math in fp32, one rounding to the input dtype.
    """
    if rotary_type not in ("llama", "hf-llama"):
        raise ValueError(f"Unknown rotary type: {rotary_type}")
    require_cuda(q, k, cos, sin)
    assert cos.dtype == torch.float32 and sin.dtype == torch.float32
    assert cos.is_contiguous() and sin.is_contiguous()
    assert q.stride(-1) == 1 and k.stride(-1) == 1
    bs = q.shape[0]
    d = q.shape[-1]
    assert cos.shape == (bs, d // 2), f"{cos.shape} {q.shape}"
    assert q.dim() == 3 and k.dim() in (2, 3)
    out_q = torch.empty(q.shape, dtype=q.dtype, device=q.device)
    out_k = torch.empty(k.shape, dtype=k.dtype, device=k.device)
    kh = 1 if k.dim() == 2 else k.shape[1]
    k_sb = k.stride(0)
    k_sh = 0 if k.dim() == 2 else k.stride(1)
    ok_sb = out_k.stride(0)
    ok_sh = 0 if k.dim() == 2 else out_k.stride(1)
    check(
        _lib.lib().chitu_hip_rope(
            ptr(q), ptr(k), ptr(out_q), ptr(out_k), ptr(cos), ptr(sin), float_dtype_code(q.dtype),
            i32(bs), i32(q.shape[1]), i32(kh), i32(d),
            i64(q.stride(0)), i64(q.stride(1)), i64(k_sb), i64(k_sh),
            i64(out_q.stride(0)), i64(out_q.stride(1)), i64(ok_sb), i64(ok_sh),
            i32(0 if rotary_type == "llama" else 1), stream_ptr(),
        ),
        "apply_rotary_pos_emb",
    )
    return out_q, out_k


# ---- ops on the decode path that the reference runs as torch glue -----------------


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6, out_bf16: bool = True, quant: str = None,
             add: torch.Tensor = None, tile_major: bool = False):
    """RMSNorm (chitu/models/model.py:29-78), optionally fused with the FP8 quantisation that the
    next fp8 linear would run on its output (model_deepseek_v3.py:98-100).

    x [..., dim] bf16 (last dim contiguous, uniform row stride); weight [dim] bf16.
    quant: None | "act" (act_quant_deepseek_v3) | "group" (per_token_group_quant_fp8, eps 1e-10) | "int8" (per-token int8,
    quantize/w8a8.py quant_act: q int8 [..., dim], s [...]; needs a plain residual `add`).
    add: optional residual branch; the kernel first forms x_new = bf16(x + add) (the reference's
    `x = x + attn(...)`, model_deepseek_v3.py:1107-1113) and normalises that.  add may carry one extra
    dim, [..., terms, dim] (terms <= 16): the terms are summed first with one bf16 rounding -- the fused
    MoE's top-k sum (fused_experts(reduce_topk=False)) folded into this launch.
    Returns y, or (y, q, s) when quant is set (y is None if out_bf16=False); with `add`, x_new is
    prepended: (x_new, y) / (x_new, y, q, s).
    tile_major (needs add and quant): q is a TiledQuant and s is None -- the same codes and scales in the layout
    the small-batch GEMM reads coalesced (fp8_gemm_deepseek_v3(TiledQuant, ...)).
    """
    require_cuda(x, weight)
    assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
    dim = x.shape[-1]
    assert x.stride(-1) == 1 and weight.is_contiguous() and weight.numel() == dim
    x2 = x.reshape(-1, dim) if x.dim() != 2 else x
    if x2.stride(-1) != 1:
        x2 = x2.contiguous()
    rows = x2.shape[0]
    a2 = sum_out = None
    terms, term_stride = 1, 0
    if add is not None:
        assert add.dtype == torch.bfloat16
        if add.dim() == x.dim() + 1:
            assert add.shape[:-2] == x.shape[:-1] and add.shape[-1] == dim and add.is_contiguous()
            terms, term_stride = add.shape[-2], dim
            a2 = add.reshape(-1, terms * dim)
        else:
            assert add.shape == x.shape
            a2 = add.reshape(-1, dim)
            if a2.stride(-1) != 1:
                a2 = a2.contiguous()
        sum_out = torch.empty(rows, dim, dtype=torch.bfloat16, device=x.device)
    y = torch.empty(rows, dim, dtype=torch.bfloat16, device=x.device) if out_bf16 else None
    q = s = None
    mode = 0
    if quant is not None:
        mode = {"act": 1, "group": 2, "int8": 3}[quant]
        if quant == "int8":
            assert add is not None and terms == 1 and not tile_major, "int8 output: the residual-add form with one term"
            q = torch.empty(rows, dim, dtype=torch.int8, device=x.device)
            s = torch.empty(rows, dtype=torch.float32, device=x.device)
        elif tile_major:
            assert add is not None and x.dim() == 2, "tile-major output: the residual-add (wide row) form on [rows, dim]"
            mode += 4
            q, s = _tiled_buffers(rows, dim, x.device)
        else:
            q = torch.empty(rows, dim, dtype=torch.float8_e4m3fn, device=x.device)
            s = torch.empty(rows, dim // 128, dtype=torch.float32, device=x.device)
    check(
        _lib.lib().chitu_hip_rmsnorm(
            ptr(x2), i64(x2.stride(0)), ptr(a2), i64(a2.stride(0) if a2 is not None else 0), i32(terms), i64(term_stride),
            ptr(sum_out), i64(dim),
            ptr(weight), ptr(y), i64(dim), i64(rows), i32(dim), f32(eps),
            ptr(q), ptr(s), i32(mode), f32(1e-10), stream_ptr(),
        ),
        "rms_norm",
    )
    if y is not None:
        y = y.view(*x.shape[:-1], dim)
    if quant is not None and tile_major:
        res = (y, TiledQuant(q, s, rows, dim), None)
    else:
        res = ((y,) if quant is None else (y, q.view(*x.shape[:-1], dim), s.view(*x.shape[:-1])) if quant == "int8"
               else (y, q.view(*x.shape[:-1], dim), s.view(*x.shape[:-1], dim // 128)))
    if add is not None:
        res = (sum_out.view(x.shape),) + res
    return res[0] if len(res) == 1 else res


def absorb_bmm_fp8(x: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, scale_offset: int, scale_stride_h: int,
                   scale_stride_n: int, scale_stride_k: int) -> torch.Tensor:
    """out[b,h,n] = sum_k x[b,h,k] * bf16(float(w[h,n,k]) * scale[...])  -- the two absorb einsums of
    MLA decode (model_deepseek_v3.py:529-531, 697) with wkv_b de-quantised in registers.

    x [B, H, K] bf16 (K contiguous); w [H, N, K] fp8 (dense rows, any head stride); scale: the flat fp32 block-scale
    tensor of wkv_b, indexed as offset + h*stride_h + (n//128)*stride_n + (k//128)*stride_k.
    """
    require_cuda(x, w, scale)
    assert x.dtype == torch.bfloat16 and w.element_size() == 1 and scale.dtype == torch.float32
    assert x.dim() == 3 and w.dim() == 3 and x.stride(-1) == 1
    assert w.stride(2) == 1 and w.stride(1) == w.shape[2], "w rows must be dense; only the head stride is free"
    B, H, K = x.shape
    assert w.shape[0] == H and w.shape[2] == K
    N = w.shape[1]
    out = torch.empty(B, H, N, dtype=torch.bfloat16, device=x.device)
    check(
        _lib.lib().chitu_hip_absorb_bmm_fp8(
            ptr(x), i64(x.stride(0)), i64(x.stride(1)), ptr(w), i64(w.stride(0)), ptr(scale), i64(scale_offset),
            i64(scale_stride_h), i64(scale_stride_n), i64(scale_stride_k), ptr(out), i64(out.stride(0)),
            i64(out.stride(1)), i32(B), i32(H), i32(N), i32(K), stream_ptr(),
        ),
        "absorb_bmm_fp8",
    )
    return out


def bf16_linear(x: torch.Tensor, weight: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """F.linear(x, weight) for bf16 weights at decode batch sizes (weight-streaming skinny GEMM)."""
    require_cuda(x, weight)
    assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
    assert x.is_contiguous() and weight.is_contiguous()
    K = x.shape[-1]
    M = x.numel() // K
    N = weight.shape[0]
    out = torch.empty(*x.shape[:-1], N, dtype=out_dtype or torch.bfloat16, device=x.device)
    check(
        _lib.lib().chitu_hip_bf16_gemm(ptr(x), ptr(weight), ptr(out), float_dtype_code(out.dtype), i64(M), i64(N),
                                       i64(K), i32(1), ptr(None), stream_ptr()),
        "bf16_linear",
    )
    return out


def gqa_qkv_post(qkv, q_heads, kv_heads, cos, sin, k_cache, v_cache, page_table, old_seq_lens, rotary_type="llama"):
    """RoPE(q in place, k) + append of the rotated k and of v to their pages in one launch (GQA / MHA
    decode).  qkv [bs, q_heads + 2*kv_heads, head_dim] bf16 (merged projection output); returns the q view."""
    require_cuda(qkv, cos, sin, k_cache, v_cache, page_table, old_seq_lens)
    assert qkv.dtype == torch.bfloat16 and qkv.dim() == 3 and qkv.is_contiguous() and qkv.shape[1] == q_heads + 2 * kv_heads
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape and k_cache.dtype == torch.bfloat16
    assert tuple(k_cache.shape[2:]) == (kv_heads, qkv.shape[2]) and page_table.dtype == torch.int32 and page_table.stride(1) == 1
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and old_seq_lens.dtype == torch.int32
    bs, _, d = qkv.shape
    assert cos.shape == (bs, d // 2) and page_table.shape[0] >= bs and page_table.is_contiguous()
    check(
        _lib.lib().chitu_hip_gqa_qkv_post(
            ptr(qkv), i64(qkv.stride(0)), i32(q_heads), i32(kv_heads), i32(d), ptr(cos), ptr(sin),
            i32(0 if rotary_type == "llama" else 1), ptr(k_cache), ptr(v_cache), i64(k_cache.shape[0]), i32(k_cache.shape[1]),
            ptr(page_table), i32(page_table.shape[1]), ptr(old_seq_lens), i32(bs), stream_ptr(),
        ),
        "gqa_qkv_post",
    )
    return qkv[:, :q_heads]


def bf16_linear_silu(x: torch.Tensor, w13: torch.Tensor) -> torch.Tensor:
    """silu(x w1^T) * (x w3^T) for w13 = [w1; w3] (bf16), one launch: == silu_and_mul(bf16_linear(x, w13))."""
    require_cuda(x, w13)
    assert x.dtype == torch.bfloat16 and w13.dtype == torch.bfloat16 and x.is_contiguous() and w13.is_contiguous()
    K = x.shape[-1]
    inter = w13.shape[0] // 2
    assert w13.shape[0] == 2 * inter and w13.shape[1] == K
    M = x.numel() // K
    if M >= 256:
        # prefill-sized M: the fused kernel streams w13 once per 32 rows (64 passes over 235 MB for a 2048-token Llama-3-8B
        # prompt: 54 % of that prefill's FLOPs at ~2 % of the MFMA peak); the compute-shaped GEMM + SiluAndMul on its bf16
        # output is the same arithmetic with the same rounding points (F.silu and the product each round to bf16, model.py:201-214)
        y = bf16_linear(x.reshape(M, K), w13)
        return (torch.nn.functional.silu(y[:, :inter]) * y[:, inter:]).view(*x.shape[:-1], inter)
    out = torch.empty(*x.shape[:-1], inter, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().chitu_hip_bf16_gemm_silu(ptr(x), ptr(w13), ptr(out), i64(M), i64(inter), i64(K), stream_ptr()),
          "bf16_linear_silu")
    return out


def bf16_add_norm_fits(M: int, N: int, K: int) -> bool:
    """Shapes chitu_hip_bf16_gemm_add_norm / chitu_hip_bf16_gemm_silu_add_norm take (N = output rows of the GEMM, or the
    SwiGLU width): 1-4 tokens, a hidden size the wide norm form covers, and a K split of at least 4 waves."""
    if not (1 <= M <= 4 and K % 64 == 0 and 512 <= K <= 8192 and M * K <= 24576):
        return False
    tiles, wk = (N + 15) // 16, 8
    while wk > 1 and (wk > K // 64 or tiles * wk > 4096):
        wk >>= 1
    return wk >= 4


def fp8_linear_add_norm_fits(M: int, N: int, K: int, terms: int = 1) -> bool:
    """Shapes chitu_hip_fp8_gemm_add_norm takes (mirrors its launcher and fp8_gemm.hip::plan_split): one row (two without a
    terms sum), K a multiple of 128 up to 8192, a GEMM that runs as one workgroup per 16 output rows with >= 4 waves."""
    if not ((M == 1 or (M == 2 and terms == 1)) and K % 128 == 0 and 128 <= K <= 8192 and 1 <= terms <= 16):
        return False
    tiles, kb = (N + 15) // 16, K // 128
    t = max(1, min(kb, (1536 + tiles - 1) // tiles))
    wk = 1
    while wk * 2 <= t and wk < 8:
        wk *= 2
    if tiles * wk < 256 and N * K >= (24 << 20):
        return False
    while wk > 1 and wk > kb:
        wk >>= 1
    return wk >= 4


def fp8_linear_add_norm(x, add, norm_weight, eps, weight, weight_scale, out_dtype=torch.bfloat16):
    """(x_new, fp8_linear(rms_norm(x_new))) with x_new = x + add [summed over its terms first], ONE launch: the residual
    add, the experts' top-k sum, attn_norm, act_quant and the wqkv_a GEMM of a batch-1 decode step
    (TransformerBlockDeepSeekV3.forward, model_deepseek_v3.py:1107-1113; linear_deepseek_v3 :98-100).  add [M, K] or
    [M, terms, K].  Bit-identical to rms_norm(x, add=add, quant="act") + fp8_gemm_deepseek_v3."""
    require_cuda(x, add, norm_weight, weight, weight_scale)
    assert x.dtype == torch.bfloat16 and add.dtype == torch.bfloat16 and x.dim() == 2 and x.stride(1) == 1
    assert weight.element_size() == 1 and weight_scale.dtype == torch.float32 and weight.is_contiguous() and weight_scale.is_contiguous()
    M, K = x.shape
    N = weight.shape[0]
    if add.dim() == 3:
        assert add.shape[0] == M and add.shape[2] == K and add.is_contiguous()
        terms, term_stride, add_stride = add.shape[1], K, add.shape[1] * K
    else:
        assert add.shape == x.shape and add.stride(1) == 1
        terms, term_stride, add_stride = 1, 0, add.stride(0)
    assert fp8_linear_add_norm_fits(M, N, K, terms)
    x_new = torch.empty(M, K, dtype=torch.bfloat16, device=x.device)
    out = torch.empty(M, N, dtype=out_dtype, device=x.device)
    check(
        _lib.lib().chitu_hip_fp8_gemm_add_norm(
            ptr(x), i64(x.stride(0)), ptr(add), i64(add_stride), i32(terms), i64(term_stride), ptr(x_new), i64(K),
            ptr(norm_weight), f32(eps), ptr(weight), ptr(weight_scale), ptr(out), float_dtype_code(out.dtype), i64(M), i64(N),
            i64(K), stream_ptr()),
        "fp8_linear_add_norm",
    )
    return x_new, out


def bf16_linear_add_norm(x, add, norm_weight, eps, weight, out_dtype=None):
    """(x_new, F.linear(rms_norm(x_new), weight)) with x_new = x + add, ONE launch (the add and the norm run as the
    GEMM's prologue in every workgroup; bit-identical to rms_norm(x, add=add) followed by bf16_linear).
    x, add [M, K] bf16 with M <= 4: check bf16_add_norm_fits first.  add may be [M, 2, K] (contiguous): the two terms are summed
    first with one bf16 rounding -- a top-2 MoE's un-summed outputs (fused_experts(reduce_topk=False)), as rms_norm(add=<3-D>)."""
    require_cuda(x, add, norm_weight, weight)
    assert x.dtype == torch.bfloat16 and add.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16
    terms = 1
    if add.dim() == 3:
        assert add.shape[0] == x.shape[0] and add.shape[1] == 2 and add.shape[2] == x.shape[1] and add.is_contiguous()
        terms = 2
    else:
        assert add.shape == x.shape and add.stride(1) == 1
    assert x.dim() == 2 and x.stride(1) == 1 and weight.is_contiguous()
    assert norm_weight.dtype == torch.bfloat16 and norm_weight.is_contiguous() and norm_weight.numel() == x.shape[1]
    M, K = x.shape
    N = weight.shape[0]
    assert weight.shape[1] == K and bf16_add_norm_fits(M, N, K), "use rms_norm(add=) + bf16_linear for this shape"
    x_new = torch.empty(M, K, dtype=torch.bfloat16, device=x.device)
    out = torch.empty(M, N, dtype=out_dtype or torch.bfloat16, device=x.device)
    check(
        _lib.lib().chitu_hip_bf16_gemm_add_norm(ptr(x), i64(x.stride(0)), ptr(add), i64(add.stride(0)), i32(terms),
                                                i64(add.stride(1) if terms == 2 else 0), ptr(x_new), i64(K),
                                                ptr(norm_weight), f32(eps), ptr(weight), ptr(out), float_dtype_code(out.dtype),
                                                i64(M), i64(N), i64(K), stream_ptr()),
        "bf16_linear_add_norm",
    )
    return x_new, out


def bf16_linear_add_norm_qkv_post(x, add, norm_weight, eps, wqkv, q_heads, kv_heads, cos, sin, k_cache, v_cache,
                                  page_table, old_seq_lens):
    """bf16_linear_add_norm on the merged q|k|v projection with gqa_qkv_post (rotary_type "llama") in its epilogue: returns
    (x_new, qkv [M, q_heads + 2*kv_heads, head_dim]) where only the q heads of qkv are written (rotated); the rotated k
    heads and the v heads are in their page rows.  Bit-identical to rms_norm(add=) + bf16_linear + gqa_qkv_post."""
    require_cuda(x, add, norm_weight, wqkv, cos, sin, k_cache, v_cache, page_table, old_seq_lens)
    assert x.dtype == torch.bfloat16 and add.dtype == torch.bfloat16 and wqkv.dtype == torch.bfloat16
    assert x.dim() == 2 and add.shape == x.shape and x.stride(1) == 1 and add.stride(1) == 1 and wqkv.is_contiguous()
    assert norm_weight.dtype == torch.bfloat16 and norm_weight.is_contiguous() and norm_weight.numel() == x.shape[1]
    M, K = x.shape
    N = wqkv.shape[0]
    d = N // (q_heads + 2 * kv_heads)
    assert N == (q_heads + 2 * kv_heads) * d and d % 16 == 0 and wqkv.shape[1] == K and bf16_add_norm_fits(M, N, K)
    assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape and k_cache.dtype == torch.bfloat16
    assert tuple(k_cache.shape[2:]) == (kv_heads, d) and page_table.dtype == torch.int32 and page_table.is_contiguous()
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape == (M, d // 2)
    assert old_seq_lens.dtype == torch.int32 and page_table.shape[0] >= M
    x_new = torch.empty(M, K, dtype=torch.bfloat16, device=x.device)
    qkv = torch.empty(M, q_heads + 2 * kv_heads, d, dtype=torch.bfloat16, device=x.device)
    check(
        _lib.lib().chitu_hip_bf16_gemm_add_norm_qkv_post(
            ptr(x), i64(x.stride(0)), ptr(add), i64(add.stride(0)), ptr(x_new), i64(K), ptr(norm_weight), f32(eps), ptr(wqkv),
            ptr(qkv), i64(M), i64(K), i32(q_heads), i32(kv_heads), i32(d), ptr(cos), ptr(sin), ptr(k_cache), ptr(v_cache),
            i64(k_cache.shape[0]), i32(k_cache.shape[1]), ptr(page_table), i32(page_table.shape[1]), ptr(old_seq_lens),
            stream_ptr()),
        "bf16_linear_add_norm_qkv_post",
    )
    return x_new, qkv


def bf16_linear_silu_add_norm(x, add, norm_weight, eps, w13):
    """(x_new, silu(y w1^T) * (y w3^T)) with x_new = x + add, y = rms_norm(x_new), ONE launch; bit-identical to
    rms_norm(x, add=add) followed by bf16_linear_silu.  Check bf16_add_norm_fits(M, inter, K) first."""
    require_cuda(x, add, norm_weight, w13)
    assert x.dtype == torch.bfloat16 and add.dtype == torch.bfloat16 and w13.dtype == torch.bfloat16
    assert x.dim() == 2 and add.shape == x.shape and x.stride(1) == 1 and add.stride(1) == 1 and w13.is_contiguous()
    assert norm_weight.dtype == torch.bfloat16 and norm_weight.is_contiguous() and norm_weight.numel() == x.shape[1]
    M, K = x.shape
    inter = w13.shape[0] // 2
    assert w13.shape == (2 * inter, K) and bf16_add_norm_fits(M, inter, K), "use rms_norm(add=) + bf16_linear_silu for this shape"
    x_new = torch.empty(M, K, dtype=torch.bfloat16, device=x.device)
    out = torch.empty(M, inter, dtype=torch.bfloat16, device=x.device)
    check(
        _lib.lib().chitu_hip_bf16_gemm_silu_add_norm(ptr(x), i64(x.stride(0)), ptr(add), i64(add.stride(0)), ptr(x_new), i64(K),
                                                     ptr(norm_weight), f32(eps), ptr(w13), ptr(out), i64(M), i64(inter), i64(K),
                                                     stream_ptr()),
        "bf16_linear_silu_add_norm",
    )
    return x_new, out


def silu_and_mul(x: torch.Tensor) -> torch.Tensor:
    """silu(x[..., :d]) * x[..., d:] on bf16 (SiluAndMul, fused_moe.py:24-39; Llama's F.silu(w1 x) * w3 x)."""
    require_cuda(x)
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and x.shape[-1] % 16 == 0
    d = x.shape[-1] // 2
    out = torch.empty(*x.shape[:-1], d, dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().chitu_hip_silu_and_mul(ptr(x), ptr(out), i64(x.numel() // (2 * d)), i64(d), stream_ptr()), "silu_and_mul")
    return out


_GATE_SPLITS = 16


_route_tickets = {}


def _route_ticket(device) -> torch.Tensor:
    """The zero-initialised word chitu_hip_gate_route_align's workgroups count themselves on (reset by
    the kernel).  Created on the first eager call per device -- before any graph capture (decode() warms up
    eagerly) -- and kept for the life of the process, so captured launches keep a valid address."""
    key = (torch.device(device).index or 0, workspace._namespace)  # launches that share a ticket must not overlap
    t = _route_tickets.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("gate_deepseek_v3(align=...) must run once eagerly before graph capture")
        t = _route_tickets[key] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def gate_deepseek_v3(x, weight, bias, n_groups, topk_groups, topk, score_func, route_scale,
                     extra_expert_id: int = -1, extra_weight: float = 1.0, extra_count: int = 1, align=None,
                     logits_partials=None):
    """GateDeepSeekV3.forward (chitu/models/model_deepseek_v3.py:810-842) in two launches:
    split-K skinny GEMM for the scores, then one fused routing kernel.  Returns (weights bf16
    [M, topk(+1)], indices int64 [M, topk(+1)]); the optional extra slot routes every token to
    `extra_expert_id` .. `extra_expert_id + extra_count - 1` with weight `extra_weight` (shared experts).

    align=(num_experts, block_size, expert_map or None): additionally run moe_align_block_size over the
    returned ids inside the routing launch (chitu_hip_gate_route_align) and return a third value
    (sorted_token_ids, expert_ids, num_tokens_post_pad) -- what fused_moe.moe_align_block_size(ids.flatten(),
    block_size, num_experts, expert_map) returns, for fused_experts(aligned=...)."""
    require_cuda(weight)
    E = weight.shape[0]
    if logits_partials is not None:
        # the score GEMM already ran in a launch in front; x is not read
        assert logits_partials.dtype == torch.float32 and logits_partials.dim() == 3 and logits_partials.is_contiguous()
        assert logits_partials.shape[2] == E
        M, dev = logits_partials.shape[1], logits_partials.device
    else:
        require_cuda(x)
        assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.is_contiguous() and weight.is_contiguous()
        M, K = x.shape
        dev = x.device
        # decode: K cut over 16 workgroups per tile, the routing launch sums the fp32 planes; prefill-sized M: the tiled GEMM,
        # its K range cut so that (128 x 128 tiles) x planes fills the 256 CUs (a 2048-token prompt against 256 experts is 32 tiles)
        if M < 256:
            splits = _GATE_SPLITS if K % (64 * _GATE_SPLITS) == 0 else 1
        else:
            tiles = ((M + 127) // 128) * ((E + 127) // 128)
            splits = 1
            while splits < 8 and tiles * splits < 256 and (K // 64) // (splits * 2) >= 4:
                splits *= 2
    cols = topk + (extra_count if extra_expert_id >= 0 else 0)
    w_out = torch.empty(M, cols, dtype=torch.bfloat16, device=dev)
    ids = torch.empty(M, cols, dtype=torch.int64, device=dev)
    lib = _lib.lib()
    if logits_partials is not None:
        logits, nparts = logits_partials, logits_partials.shape[0]
    elif splits > 1:
        part = torch.empty(splits, M, E, dtype=torch.float32, device=x.device)
        check(lib.chitu_hip_bf16_gemm(ptr(x), ptr(weight), ptr(None), i32(0), i64(M), i64(E), i64(K), i32(splits),
                                      ptr(part), stream_ptr()), "gate scores")
        logits, nparts = part, splits
    else:
        logits = bf16_linear(x, weight)
        nparts = 0
    sf = {"softmax": 0, "sigmoid": 1, "softmax_renorm": 2}[score_func]
    if align is None:
        check(
            lib.chitu_hip_gate_route(ptr(logits), i32(nparts), i64(M), i32(E), ptr(bias), i32(n_groups), i32(topk_groups),
                                     i32(topk), i32(sf), f32(route_scale), ptr(w_out), ptr(ids), i32(cols),
                                     i32(extra_expert_id), f32(extra_weight), i32(extra_count), stream_ptr()),
            "gate_route",
        )
        return w_out, ids
    a_experts, a_block, a_map = align
    cap = M * cols + a_experts * (a_block - 1)
    nblk = (cap + a_block - 1) // a_block
    sorted_ids = torch.empty(cap, dtype=torch.int32, device=dev)
    expert_ids = torch.empty(nblk, dtype=torch.int32, device=dev)
    npost = torch.empty(1, dtype=torch.int32, device=dev)
    cumsum = torch.empty(a_experts + 1, dtype=torch.int32, device=dev)
    if a_map is not None:
        assert a_map.dtype == torch.int32 and a_map.is_cuda and a_map.is_contiguous() and a_map.numel() >= a_experts
    check(
        lib.chitu_hip_gate_route_align(ptr(logits), i32(nparts), i64(M), i32(E), ptr(bias), i32(n_groups), i32(topk_groups),
                                       i32(topk), i32(sf), f32(route_scale), ptr(w_out), ptr(ids), i32(cols),
                                       i32(extra_expert_id), f32(extra_weight), i32(extra_count), i32(a_experts),
                                       i32(a_block), ptr(sorted_ids), i64(cap), ptr(expert_ids), i64(nblk), ptr(npost),
                                       ptr(cumsum), ptr(a_map), ptr(_route_ticket(dev)), stream_ptr()),
        "gate_route_align",
    )
    return w_out, ids, (sorted_ids, expert_ids, npost)


def mla_kv_prep(kv_in, q_pe, cos, sin, kv_norm_weight, eps, kv_cache, page_table, old_seq_lens):
    """kv_norm + RoPE(q_pe in place, k_pe) + paged append in one launch (see chitu_hip_mla_kv_prep).
    kv_in [bs, 576] (row-strided view of wqkv_a's output), q_pe [bs, H, 64] view, rotated in place."""
    require_cuda(kv_in, q_pe, cos, sin, kv_norm_weight, kv_cache, page_table, old_seq_lens)
    assert kv_in.dtype == torch.bfloat16 and q_pe.dtype == torch.bfloat16 and kv_cache.dtype == torch.bfloat16
    assert kv_in.shape[-1] == 576 and kv_in.stride(-1) == 1 and q_pe.stride(-1) == 1 and q_pe.shape[-1] == 64
    assert kv_cache.is_contiguous() and kv_cache.shape[-1] == 576 and page_table.is_contiguous()
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
    bs = kv_in.shape[0]
    check(
        _lib.lib().chitu_hip_mla_kv_prep(
            ptr(kv_in), i64(kv_in.stride(0)), ptr(q_pe), i64(q_pe.stride(0)), i64(q_pe.stride(1)), i32(q_pe.shape[1]),
            ptr(cos), ptr(sin), ptr(kv_norm_weight), f32(eps), ptr(kv_cache), i64(kv_cache.shape[0]),
            i32(kv_cache.shape[1]), ptr(page_table), i32(page_table.shape[1]), ptr(old_seq_lens), i32(bs), i32(512),
            i32(64), stream_ptr(),
        ),
        "mla_kv_prep",
    )


def mla_merge_absorb_uv_quant_fp8(partials, num_splits, batch, w, scale, scale_offset, scale_stride_h, scale_stride_k,
                                  tile_major: bool = False):
    """Split-KV merge of the MLA partials + absorb_uv_quant_fp8 in one launch (one workgroup per
    (head, token), meant for decode batches up to a few dozen tokens).  partials: the workspace
    returned by HipAttnBackend.mla_decode(return_partials=True)."""
    require_cuda(partials, w, scale)
    assert w.element_size() == 1 and scale.dtype == torch.float32 and w.dim() == 3 and w.shape[1] == 128
    assert w.stride(2) == 1 and w.stride(1) == w.shape[2] and w.shape[2] == 512 and num_splits >= 2
    H = w.shape[0]
    assert partials.numel() * partials.element_size() >= batch * H * num_splits * (512 * 2 + 4)  # bf16 rows + fp32 LSE
    if tile_major:  # (TiledQuant, None): the fp8 row of every token tile-major, for fp8_gemm_deepseek_v3(TiledQuant, ...)
        q, s = _tiled_buffers(batch, H * 128, w.device)
        fn = _lib.lib().chitu_hip_mla_merge_absorb_uv_quant_fp8_tm
    else:
        q = torch.empty(batch, H * 128, dtype=torch.float8_e4m3fn, device=w.device)
        s = torch.empty(batch, H, dtype=torch.float32, device=w.device)
        fn = _lib.lib().chitu_hip_mla_merge_absorb_uv_quant_fp8
    check(
        fn(ptr(partials), i32(num_splits), ptr(w), i64(w.stride(0)), ptr(scale), i64(scale_offset),
           i64(scale_stride_h), i64(scale_stride_k), ptr(q), ptr(s), i32(batch), i32(H), i32(512), stream_ptr()),
        "mla_merge_absorb_uv_quant_fp8",
    )
    return (TiledQuant(q, s, batch, H * 128), None) if tile_major else (q, s)


def embed_rope_gather(tokens, embed_weight, vocab_start, positions=None, cos_table=None, sin_table=None):
    """The decode step's prologue in one launch: h = vocabulary-parallel embedding rows (zero for ids outside
    [vocab_start, vocab_start + rows), tensor_parallel.py:199-208 without the all-reduce) and, when positions is
    given, (cos, sin) = the rotary rows of positions[:bs] (model.py:429-448).  Returns (h, cos, sin)."""
    require_cuda(tokens, embed_weight, positions, cos_table, sin_table)
    assert tokens.dtype == torch.int64 and tokens.dim() == 1 and tokens.is_contiguous()
    assert embed_weight.dtype == torch.bfloat16 and embed_weight.is_contiguous()
    bs, dim = tokens.shape[0], embed_weight.shape[1]
    h = torch.empty(bs, dim, dtype=torch.bfloat16, device=tokens.device)
    cos = sin = None
    half = rows = 0
    if positions is not None:
        assert positions.dtype == torch.int32 and positions.is_contiguous() and positions.numel() >= bs
        assert cos_table.dtype == torch.float32 and cos_table.is_contiguous() and sin_table.is_contiguous()
        rows, half = cos_table.shape
        cos = torch.empty(bs, half, dtype=torch.float32, device=tokens.device)
        sin = torch.empty(bs, half, dtype=torch.float32, device=tokens.device)
    check(
        _lib.lib().chitu_hip_embed_rope_gather(ptr(tokens), ptr(embed_weight), i64(vocab_start), i64(embed_weight.shape[0]),
                                               i32(dim), ptr(h), ptr(positions), ptr(cos_table), ptr(sin_table), i64(rows),
                                               i32(half), ptr(cos), ptr(sin), i32(bs), stream_ptr()),
        "embed_rope_gather",
    )
    return h, cos, sin


def mla_qkv_post(q_a_kv, q_lora_rank, q_norm_weight, q_eps, kv_norm_weight, kv_eps, cos, sin, kv_cache, page_table,
                 old_seq_lens):
    """One launch for everything that reads wqkv_a's output [bs, q_lora + 512 + 64]: q_norm + act_quant
    (returns (q_fp8, q_scales), the wq_b GEMM input) and kv_norm + RoPE(k_pe) + page append.  q_pe is
    rotated later by absorb_bmm_rope_fp8.  q_a_kv: the bf16 GEMM output, or the fp32 split-K planes
    [S, bs, q_lora + 576] of fp8_gemm_partials_deepseek_v3 (summed in plane order, rounded once, as they load)."""
    require_cuda(q_a_kv, q_norm_weight, kv_norm_weight, cos, sin, kv_cache, page_table, old_seq_lens)
    if q_a_kv.dtype == torch.float32:
        assert q_a_kv.dim() == 3 and q_a_kv.is_contiguous()
        nparts, bs, stride = q_a_kv.shape[0], q_a_kv.shape[1], q_a_kv.shape[2]
    else:
        assert q_a_kv.dtype == torch.bfloat16 and q_a_kv.dim() == 2 and q_a_kv.stride(1) == 1
        nparts, bs, stride = 0, q_a_kv.shape[0], q_a_kv.stride(0)
    assert q_a_kv.shape[-1] == q_lora_rank + 576 and kv_cache.dtype == torch.bfloat16 and kv_cache.shape[-1] == 576
    assert kv_cache.is_contiguous() and page_table.is_contiguous() and page_table.dtype == torch.int32
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and old_seq_lens.dtype == torch.int32
    q = torch.empty(bs, q_lora_rank, dtype=torch.float8_e4m3fn, device=q_a_kv.device)
    s = torch.empty(bs, q_lora_rank // 128, dtype=torch.float32, device=q_a_kv.device)
    check(
        _lib.lib().chitu_hip_mla_qkv_post(
            ptr(q_a_kv), i32(nparts), i64(stride), i32(q_lora_rank), ptr(q_norm_weight), f32(q_eps), ptr(q), ptr(s),
            ptr(kv_norm_weight), f32(kv_eps), ptr(cos), ptr(sin), ptr(kv_cache), i64(kv_cache.shape[0]),
            i32(kv_cache.shape[1]), ptr(page_table), i32(page_table.shape[1]), ptr(old_seq_lens), i32(bs), i32(512),
            i32(64), stream_ptr(),
        ),
        "mla_qkv_post",
    )
    return q, s


def mla_q_proj_fits(bs: int, q_lora_rank: int) -> bool:
    """Shapes chitu_hip_mla_q_proj takes (decode batches up to two 16-token tiles, q_lora_rank <= 8 waves x 2 K blocks)."""
    return 0 < bs <= 32 and q_lora_rank % 128 == 0 and 128 <= q_lora_rank <= 2048


def mla_q_proj(q_a_kv, q_lora_rank, q_norm_weight, q_eps, wq_b, wq_b_scale, kv_norm_weight, kv_eps, cos, sin, kv_cache,
               page_table, old_seq_lens, out_dtype=None):
    """mla_qkv_post + the wq_b GEMM in ONE launch: q = fp8_gemm(act_quant(q_norm(q_a)), wq_b) [bs, N] with the norm and
    the quantisation as the GEMM's prologue (model_deepseek_v3.py:488), and each token's [kv_norm(kv_c) | RoPE(k_pe)] row
    appended to its page (:493-496, :684-686) by extra workgroups of the same grid.  q_a_kv: wqkv_a's bf16 output
    [bs, q_lora + 576]; check mla_q_proj_fits first."""
    require_cuda(q_a_kv, q_norm_weight, wq_b, wq_b_scale, kv_norm_weight, cos, sin, kv_cache, page_table, old_seq_lens)
    assert q_a_kv.dtype == torch.bfloat16 and q_a_kv.dim() == 2 and q_a_kv.stride(1) == 1
    bs = q_a_kv.shape[0]
    assert mla_q_proj_fits(bs, q_lora_rank), "use mla_qkv_post + fp8_gemm_deepseek_v3 for this shape"
    assert q_a_kv.shape[-1] == q_lora_rank + 576 and kv_cache.dtype == torch.bfloat16 and kv_cache.shape[-1] == 576
    assert kv_cache.is_contiguous() and page_table.is_contiguous() and page_table.dtype == torch.int32
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous() and old_seq_lens.dtype == torch.int32
    assert wq_b.element_size() == 1 and wq_b.is_contiguous() and wq_b.shape[1] == q_lora_rank
    assert wq_b_scale.dtype == torch.float32 and wq_b_scale.is_contiguous()
    N = wq_b.shape[0]
    assert wq_b_scale.shape == ((N + 127) // 128, q_lora_rank // 128)
    out = torch.empty(bs, N, dtype=out_dtype or torch.get_default_dtype(), device=q_a_kv.device)
    check(
        _lib.lib().chitu_hip_mla_q_proj(
            ptr(q_a_kv), i64(q_a_kv.stride(0)), i32(q_lora_rank), ptr(q_norm_weight), f32(q_eps), ptr(wq_b), ptr(wq_b_scale),
            ptr(out), float_dtype_code(out.dtype), i64(N), ptr(kv_norm_weight), f32(kv_eps), ptr(cos), ptr(sin),
            ptr(kv_cache), i64(kv_cache.shape[0]), i32(kv_cache.shape[1]), ptr(page_table), i32(page_table.shape[1]),
            ptr(old_seq_lens), i32(bs), i32(512), i32(64), stream_ptr(),
        ),
        "mla_q_proj",
    )
    return out


def absorb_bmm_rope_fp8(x, w, scale, scale_offset, scale_stride_h, scale_stride_n, scale_stride_k, q_pe, cos, sin):
    """absorb_bmm_fp8 + in-place RoPE of q_pe [B, H, 64] in the same launch."""
    require_cuda(x, w, scale, q_pe, cos, sin)
    assert x.dtype == torch.bfloat16 and w.element_size() == 1 and scale.dtype == torch.float32
    assert x.dim() == 3 and w.dim() == 3 and x.stride(-1) == 1 and w.stride(2) == 1 and w.stride(1) == w.shape[2]
    assert q_pe.dtype == torch.bfloat16 and q_pe.shape[-1] == 64 and q_pe.stride(-1) == 1
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
    B, H, K = x.shape
    N = w.shape[1]
    out = torch.empty(B, H, N, dtype=torch.bfloat16, device=x.device)
    check(
        _lib.lib().chitu_hip_absorb_bmm_rope_fp8(
            ptr(x), i64(x.stride(0)), i64(x.stride(1)), ptr(w), i64(w.stride(0)), ptr(scale), i64(scale_offset),
            i64(scale_stride_h), i64(scale_stride_n), i64(scale_stride_k), ptr(out), i64(out.stride(0)),
            i64(out.stride(1)), i32(B), i32(H), i32(N), i32(K), ptr(q_pe), i64(q_pe.stride(0)), i64(q_pe.stride(1)),
            ptr(cos), ptr(sin), i32(64), stream_ptr(),
        ),
        "absorb_bmm_rope_fp8",
    )
    return out


def absorb_bmm_rope_kv_fp8(x, w, scale, scale_offset, scale_stride_h, scale_stride_n, scale_stride_k, q_pe, cos, sin,
                           kv_in, kv_norm_weight, eps, kv_cache, page_table, old_seq_lens):
    """absorb_bmm_rope_fp8 + the KV half of mla_kv_prep (kv_norm, RoPE(k_pe), page append of kv_in [bs, 576]) in the
    same launch: the decode path of models without a q low-rank projection (DeepSeek-V2-Lite), where no wq_b launch
    exists to carry the KV row."""
    require_cuda(x, w, scale, q_pe, cos, sin, kv_in, kv_norm_weight, kv_cache, page_table, old_seq_lens)
    assert x.dtype == torch.bfloat16 and w.element_size() == 1 and scale.dtype == torch.float32
    assert x.dim() == 3 and w.dim() == 3 and x.stride(-1) == 1 and w.stride(2) == 1 and w.stride(1) == w.shape[2]
    assert q_pe.dtype == torch.bfloat16 and q_pe.shape[-1] == 64 and q_pe.stride(-1) == 1
    assert cos.dtype == torch.float32 and cos.is_contiguous() and sin.is_contiguous()
    assert kv_in.dtype == torch.bfloat16 and kv_cache.dtype == torch.bfloat16 and kv_in.shape[-1] == 576 and kv_in.stride(-1) == 1
    assert kv_cache.is_contiguous() and kv_cache.shape[-1] == 576 and page_table.is_contiguous()
    B, H, K = x.shape
    N = w.shape[1]
    assert kv_in.shape[0] == B and N % 16 == 0
    out = torch.empty(B, H, N, dtype=torch.bfloat16, device=x.device)
    check(
        _lib.lib().chitu_hip_absorb_bmm_rope_kv_fp8(
            ptr(x), i64(x.stride(0)), i64(x.stride(1)), ptr(w), i64(w.stride(0)), ptr(scale), i64(scale_offset),
            i64(scale_stride_h), i64(scale_stride_n), i64(scale_stride_k), ptr(out), i64(out.stride(0)),
            i64(out.stride(1)), i32(B), i32(H), i32(N), i32(K), ptr(q_pe), i64(q_pe.stride(0)), i64(q_pe.stride(1)),
            ptr(cos), ptr(sin), i32(64), ptr(kv_in), i64(kv_in.stride(0)), ptr(kv_norm_weight), f32(eps), ptr(kv_cache),
            i64(kv_cache.shape[0]), i32(kv_cache.shape[1]), ptr(page_table), i32(page_table.shape[1]), ptr(old_seq_lens),
            i32(512), stream_ptr(),
        ),
        "absorb_bmm_rope_kv_fp8",
    )
    return out


def absorb_uv_quant_fp8(x, w, scale, scale_offset, scale_stride_h, scale_stride_k):
    """absorb_bmm_fp8 for the W_UV half (N = 128) + act_quant of its bf16 result: returns
    (q [B, H*128] e4m3fn, s [B, H] f32), the input of the wo fp8 GEMM."""
    require_cuda(x, w, scale)
    assert x.dtype == torch.bfloat16 and w.element_size() == 1 and scale.dtype == torch.float32
    assert x.dim() == 3 and w.dim() == 3 and x.stride(-1) == 1 and w.shape[1] == 128
    assert w.stride(2) == 1 and w.stride(1) == w.shape[2]
    B, H, K = x.shape
    q = torch.empty(B, H * 128, dtype=torch.float8_e4m3fn, device=x.device)
    s = torch.empty(B, H, dtype=torch.float32, device=x.device)
    check(
        _lib.lib().chitu_hip_absorb_uv_quant_fp8(
            ptr(x), i64(x.stride(0)), i64(x.stride(1)), ptr(w), i64(w.stride(0)), ptr(scale), i64(scale_offset),
            i64(scale_stride_h), i64(scale_stride_k), ptr(q), ptr(s), i32(B), i32(H), i32(K), stream_ptr(),
        ),
        "absorb_uv_quant_fp8",
    )
    return q, s
