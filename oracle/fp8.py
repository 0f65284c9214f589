"""Oracle for the FP8 quant / dequant / GEMM ops (torch CPU, fp32 math).  TEST INFRASTRUCTURE ONLY.

Each function restates one reference kernel; citations are into /root/reference.
fp8 tensors are torch.float8_e4m3fn (OCP), the encoding gfx950's MFMA consumes natively.
"""

import numpy as np
import torch

FP8_MAX = 448.0
BLOCK = 128


# Rounding hooks.  Defaults are IEEE round-to-nearest-even, i.e. what the reference does on a
# GPU (PTX cvt.rn) and what gfx950 does.  tests/test_oracle_golden.py swaps in the Triton
# *interpreter's* casts (round-half-up without exponent carry for fp8, truncation for bf16) to
# pin these restatements bit-exactly against fixtures generated under TRITON_INTERPRET=1.
def _to_fp8_rne(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.float8_e4m3fn)


def _to_out_rne(t: torch.Tensor, dtype) -> torch.Tensor:
    return t.to(dtype)


CAST = {"fp8": _to_fp8_rne, "out": _to_out_rne}


def to_fp8(t):
    return CAST["fp8"](t)


def to_out(t, dtype):
    return CAST["out"](t, dtype)


def act_quant_deepseek_v3(x: torch.Tensor, block_size: int = BLOCK):
    """chitu/triton_kernels.py:193-214: s = max|x|/448 (no eps), y = (x/s) -> e4m3fn (no clamp)."""
    shape = x.shape
    xf = x.float().reshape(-1, block_size)
    s = xf.abs().amax(dim=-1) / np.float32(FP8_MAX)
    y = xf / s[:, None]
    q = to_fp8(y).reshape(shape)
    return q, s.reshape(*shape[:-1], shape[-1] // block_size)


def per_token_group_quant_fp8(x: torch.Tensor, group_size: int = BLOCK, eps: float = 1e-10):
    """chitu/fused_moe.py:670-710: s = max(max|x|, eps)/448, q = clamp(x/s, -448, 448)."""
    shape = x.shape
    xf = x.float().reshape(-1, group_size)
    amax = torch.clamp(xf.abs().amax(dim=-1), min=eps)
    s = amax / np.float32(FP8_MAX)
    y = torch.clamp(xf / s[:, None], -FP8_MAX, FP8_MAX)
    q = to_fp8(y).reshape(shape)
    return q, s.reshape(*shape[:-1], shape[-1] // group_size)


def _expand_block_scale(s: torch.Tensor, rows: int, cols: int, block: int = BLOCK):
    return s.repeat_interleave(block, dim=-2)[..., :rows, :].repeat_interleave(block, dim=-1)[..., :cols]


def weight_dequant_deepseek_v3(w: torch.Tensor, s: torch.Tensor, out_dtype=torch.bfloat16):
    """chitu/triton_kernels.py:217-247: y = float(x) * s[m//128, n//128] -> out dtype."""
    rows, cols = w.shape[-2], w.shape[-1]
    return to_out(w.float() * _expand_block_scale(s, rows, cols), out_dtype)


def soft_decode_fp8(w: torch.Tensor) -> torch.Tensor:
    """chitu/triton_kernels.py:261: bits = ((b & 0x80) << 24) | ((b & 0x7F) << 20), as f32."""
    b = w.contiguous().view(torch.uint8).numpy().astype(np.uint32)
    bits = ((b & 0x80) << 24) | ((b & 0x7F) << 20)
    return torch.from_numpy(bits.view(np.float32).copy())


SOFT_SCALE = np.frombuffer(np.uint32(0x7B800000).tobytes(), dtype=np.float32)[0]  # 2**120


def weight_dequant_soft_fp8_deepseek_v3(w: torch.Tensor, s: torch.Tensor, out_dtype=torch.bfloat16):
    """chitu/triton_kernels.py:265-287 (step 2): y = x_bits * (s * 2^120) -> out dtype."""
    rows, cols = w.shape[-2], w.shape[-1]
    s2 = _expand_block_scale(s, rows, cols) * SOFT_SCALE
    return to_out(soft_decode_fp8(w) * s2, out_dtype)


def fp8_gemm_deepseek_v3(a_q, a_s, b_q, b_s, out_dtype=torch.bfloat16):
    """chitu/triton_kernels.py:302-365: acc += dot(a_kb, b_kb) * a_s[:, kb] * b_s[n//128, kb]."""
    K = a_q.shape[-1]
    a = a_q.float().reshape(-1, K)
    a_s = a_s.reshape(a.shape[0], -1)
    b = b_q.float()
    N = b.shape[0]
    acc = torch.zeros(a.shape[0], N, dtype=torch.float32)
    bs_rows = b_s.repeat_interleave(BLOCK, dim=0)[:N]  # [N, K/128]
    for kb in range(K // BLOCK):
        sl = slice(kb * BLOCK, (kb + 1) * BLOCK)
        dot = a[:, sl] @ b[:, sl].T
        acc += dot * a_s[:, kb : kb + 1] * bs_rows[:, kb][None, :]
    return to_out(acc, out_dtype).reshape(*a_q.shape[:-1], N)


def soft_fp8_gemm_deepseek_v3(a, b_q, b_s, out_dtype=torch.bfloat16):
    """chitu/triton_kernels.py:388-508: b' = bf16(bits(b) * (b_s * 2^120)); acc += dot(a, b')."""
    K = a.shape[-1]
    af = a.float().reshape(-1, K)
    N = b_q.shape[0]
    b_new = weight_dequant_soft_fp8_deepseek_v3(b_q, b_s, torch.bfloat16).float()
    acc = torch.zeros(af.shape[0], N, dtype=torch.float32)
    for kb in range(K // BLOCK):
        sl = slice(kb * BLOCK, (kb + 1) * BLOCK)
        acc += af[:, sl] @ b_new[:, sl].T
    return to_out(acc, out_dtype).reshape(*a.shape[:-1], N)


def linear_deepseek_v3(x, weight, weight_scale, out_dtype=torch.bfloat16):
    """chitu/models/model_deepseek_v3.py:53-106, W8A8 branch: act_quant + fp8_gemm."""
    shape = x.shape
    xq, xs = act_quant_deepseek_v3(x.reshape(-1, shape[-1]).contiguous())
    y = fp8_gemm_deepseek_v3(xq, xs, weight, weight_scale, out_dtype)
    return y.reshape(*shape[:-1], y.shape[-1])
