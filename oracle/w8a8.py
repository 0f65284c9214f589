"""Oracle for the INT8 W8A8 linear (torch CPU).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the arithmetic lives in the closed, un-vendored `w8a8gemm` / `w8a8gemv` packages
(third_party/nv_w8a8_kernels/README.md:1, no version pin).  This restates what the call sites and the
reference's test fix: quant_act / quant_weight (chitu/quantize/w8a8.py:18-35, verbatim), and
out = (q_x . q_w^T) * s_act * s_w (+ bias) with an exact integer dot (test/pytest/test_w8a8.py:13-48
compares against fp16 torch.mm at rtol=atol=5e-3 with unit scales).
"""

import torch


def quant_act(act):
    shape = act.shape
    scales = act.abs().max(dim=-1, keepdim=True)[0].to(torch.float)
    scales.clamp_(min=1e-5).div_(127.0)
    aa = act.div(scales).round_()
    return aa.to(torch.int8).view(-1, shape[-1]), scales.view(-1)


def quant_weight(w):
    scales = w.abs().max(dim=-1, keepdim=True)[0].to(torch.float)
    scales.clamp_(min=1e-5).div_(127.0)
    ww = w.div(scales).round_()
    return ww.to(torch.int8), scales.view(-1)


def w8a8_linear(q_x, act_scale, weight, scale_channel, bias=None, out_dtype=torch.float16):
    acc = q_x.to(torch.int64) @ weight.to(torch.int64).T  # exact
    out = (acc.to(torch.float32) * act_scale[:, None]) * scale_channel[None, :]
    if bias is not None:
        out = out + bias.float()
    return out.to(out_dtype)
