"""INT8 W8A8 linear (per-token activations x per-output-channel weights) on HIP.

Mirrors chitu/quantize/w8a8.py:18-164: `quant_act`, `quant_weight`, `W8A8Linear` (same buffers:
`weight` int8 [out, in], `scale_channel` f32 [out], optional fp16 `bias`), `from_float`.  The reference
forwards to the closed `w8a8gemm.mm` / `w8a8gemv.mv` (the latter when batch <= 4, :117-120); here both
cases are one weight-streaming kernel (chitu_hip_w8a8_int8_gemm) after chitu_hip_quant_act_int8.
"""

import torch
from torch import nn

from .. import _lib
from .._lib import check, float_dtype_code, i32, i64, ptr, require_cuda, stream_ptr


@torch.no_grad()
def quant_act(act: torch.Tensor):
    """(int8 [rows, K], f32 scales [rows]): s = clamp(max|row|, 1e-5)/127, q = round(x/s)  (w8a8.py:18-26)."""
    require_cuda(act)
    K = act.shape[-1]
    x = act.reshape(-1, K)
    if not x.is_contiguous():
        x = x.contiguous()
    q = torch.empty(x.shape, dtype=torch.int8, device=act.device)
    s = torch.empty(x.shape[0], dtype=torch.float32, device=act.device)
    check(
        _lib.lib().chitu_hip_quant_act_int8(ptr(x), float_dtype_code(x.dtype), i64(x.shape[0]), i64(K), ptr(q), ptr(s),
                                            stream_ptr()),
        "quant_act",
    )
    return q, s


@torch.no_grad()
def quant_weight(w: torch.Tensor):
    """Per-output-channel int8 weights (w8a8.py:29-35).  Load-time only: plain torch ops."""
    scales = w.abs().max(dim=-1, keepdim=True)[0].to(torch.float)
    scales.clamp_(min=1e-5).div_(127.0)
    ww = w.div(scales).round_()
    return ww.to(torch.int8), scales.view(-1)


def w8a8_linear(q_x, act_scale, weight, scale_channel, bias=None, out_dtype=torch.float16):
    """out[m][n] = (sum_k q_x[m][k] * weight[n][k]) * act_scale[m] * scale_channel[n] (+ bias[n])."""
    require_cuda(q_x, act_scale, weight, scale_channel)
    assert q_x.dtype == torch.int8 and weight.dtype == torch.int8
    assert q_x.is_contiguous() and weight.is_contiguous() and act_scale.dtype == torch.float32
    assert scale_channel.dtype == torch.float32 and scale_channel.is_contiguous()
    M, K = q_x.shape
    N = weight.shape[0]
    out = torch.empty(M, N, dtype=out_dtype, device=q_x.device)
    check(
        _lib.lib().chitu_hip_w8a8_int8_gemm(
            ptr(q_x), ptr(act_scale), ptr(weight), ptr(scale_channel), ptr(bias),
            i32(float_dtype_code(bias.dtype) if bias is not None else 0), ptr(out), i32(float_dtype_code(out_dtype)),
            i64(M), i64(N), i64(K), stream_ptr(),
        ),
        "w8a8_linear",
    )
    return out


class W8A8Linear(nn.Module):
    def __init__(self, in_features, out_features, bias=True, quantize_output=False, pre_norm=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.pre_norm = pre_norm
        self.register_buffer("weight", torch.zeros(out_features, in_features, dtype=torch.int8, requires_grad=False))
        self.register_buffer("scale_channel", torch.ones([out_features], dtype=torch.float, requires_grad=False))
        if bias:
            self.register_buffer("bias", torch.zeros((out_features,), dtype=torch.float16, requires_grad=False))
        else:
            self.register_buffer("bias", None)
        self.act_quant_name = "per_token"
        self.act_quant = quant_act
        assert not quantize_output, "output re-quantisation is unused on the decode path"

    @torch.no_grad()
    def forward(self, x):
        """x [tokens, in] or [bs, seq, in] -> same leading dims, `out_features` last (w8a8.py:97-132);
        output dtype fp16 like the reference's kernels, or bf16 if the input is bf16."""
        lead = x.shape[:-1]
        q_x, act_scale = self.act_quant(x)
        out_dtype = torch.bfloat16 if x.dtype == torch.bfloat16 else torch.float16
        out = w8a8_linear(q_x, act_scale, self.weight, self.scale_channel, self.bias, out_dtype)
        return out.view(*lead, self.out_features)

    @staticmethod
    def from_float(module, weight_quant="per_channel", act_quant="per_token", quantize_output=False,
                   model_arch_only=False):
        new_module = W8A8Linear(module.in_features, module.out_features, module.bias is not None,
                                quantize_output=quantize_output)
        if not model_arch_only:
            ww, scl = quant_weight(module.weight.data)
            new_module.weight = ww.contiguous()
            new_module.scale_channel = scl.contiguous()
            if module.bias is not None:
                new_module.bias = module.bias.data.to(torch.float16)
        return new_module.to(module.weight.device)

    def __repr__(self):
        return f"W8A8Linear({self.in_features}, {self.out_features}, bias={self.bias is not None})"


def replace_with_simple_w8a8(model: nn.Module, skip=("head", "lm_head", "gate"), model_arch_only=False):
    """Swap every nn.Linear (and linear-like module with weight/in_features/out_features) for a
    W8A8Linear, like replace_with_simple_w8a8 (chitu/quantize/quantizer.py:117-145)."""
    for name, child in list(model.named_children()):
        if isinstance(child, nn.Linear) and name not in skip and child.in_features % 128 == 0:
            setattr(model, name, W8A8Linear.from_float(child, model_arch_only=model_arch_only))
        else:
            replace_with_simple_w8a8(child, skip, model_arch_only)
    return model
