"""Mixtral-family decode step (GQA attention + sparse MoE) with INT8 W8A8 experts -- BASELINE config 4
("Mixtral-8x7B W8A8 fused_moe grouped-GEMM + moe_align").

Reference (read-only): chitu/models/model_hf_mixtral.py (SparseMoeBlockHFMixtral:22-94: bf16 router,
softmax -> top-2 -> renormalise, a Python loop over the experts) on chitu/models/model_hf_llama.py blocks
(rotary_type "hf-llama"), with `simple_w8a8` turning the expert linears into W8A8Linear
(chitu/quantize/quantizer.py:117-145).  The reference has no fused int8 MoE (SURVEY gap G2); here the
expert loop is fused_moe.fused_experts(use_int8_w8a8=True).  Attention, norms and the router stay bf16.
"""

from dataclasses import dataclass

import torch

from . import fused_moe, ops
from . import tensor_parallel as tp
from .deepseek_v3 import _route_align_enabled
from .llama import LlamaAttention, LlamaDecoder, _fuses_norm, _param


@dataclass
class MixtralArgs:
    """Fields of chitu/config/models/Mixtral-8x7B-Instruct-v0.1.yaml:6-17 (defaults = Mixtral-8x7B)."""

    dim: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    n_kv_heads: int = 8
    vocab_size: int = 32000
    ffn_dim: int = 14336  # intermediate_dim of every expert
    norm_eps: float = 1e-5
    rope_theta: float = 1000000.0
    num_local_experts: int = 8
    num_experts_per_tok: int = 2

    @property
    def head_dim(self):
        return self.dim // self.n_heads


# The decode step's launch fusions of round 6 (routing + align in one launch, attn_norm in the qkv GEMM's prologue, the experts'
# int8 quantisation inside ffn_norm, the top-2 sum inside the next residual add): every one bit-identical to the separate
# launches; False = the separate launches (the cross-check of tests/test_gpu_mixtral.py, not a product switch).
FUSE = True


class MixtralSparseMoe(torch.nn.Module):
    """Router (bf16) + INT8 W8A8 experts, every rank holding all experts at 1/tp of their width."""

    def __init__(self, args: MixtralArgs, device=None):
        super().__init__()
        t = tp.get_tp_size()
        self.E, self.topk, self.inter = args.num_local_experts, args.num_experts_per_tok, args.ffn_dim // t
        self.gate = _param(self.E, args.dim, device=device)
        self.w13 = torch.nn.Parameter(torch.empty(self.E, 2 * self.inter, args.dim, dtype=torch.int8, device=device), requires_grad=False)
        self.w13_scale = torch.nn.Parameter(torch.empty(self.E, 2 * self.inter, dtype=torch.float32, device=device), requires_grad=False)
        self.w2 = torch.nn.Parameter(torch.empty(self.E, args.dim, self.inter, dtype=torch.int8, device=device), requires_grad=False)
        self.w2_scale = torch.nn.Parameter(torch.empty(self.E, args.dim, dtype=torch.float32, device=device), requires_grad=False)

    def forward(self, x, x_quant=None, defer_sum: bool = False):
        """x: ffn_norm output bf16 [tokens, dim]; x_quant: its per-token int8 form when the norm launch produced it.
        defer_sum: return the un-summed [tokens, topk, dim] expert outputs; the consumer's residual add folds the top-k sum in
        (ops.rms_norm(add=<3-D>) / ops.bf16_linear_add_norm(add=<[M, 2, K]>), chitu_hip_moe_sum's arithmetic)."""
        # decode-sized batches: moe_align runs inside the routing launch (round 6; as the DeepSeek MoE does since round 1)
        align = (self.E, fused_moe._MOE_BLOCK_M, None) if FUSE and _route_align_enabled(x.shape[0]) else None
        routed = ops.gate_deepseek_v3(x, self.gate, None, 1, 1, self.topk, "softmax_renorm", 1.0, align=align)
        return fused_moe.fused_experts(x, self.w13, self.w2, routed[0], routed[1], inplace=True, use_int8_w8a8=True,
                                       w1_scale=self.w13_scale, w2_scale=self.w2_scale, global_num_experts=self.E,
                                       aligned=routed[2] if len(routed) > 2 else None, a1_quant=x_quant,
                                       reduce_topk=not defer_sum)


class MixtralBlock(torch.nn.Module):
    def __init__(self, layer_id, args, cache, attn_backend, device=None):
        super().__init__()
        self.attn = LlamaAttention(args, layer_id, cache, attn_backend, device, rotary_type="hf-llama")
        self.ffn = MixtralSparseMoe(args, device)
        self.attn_norm = _param(args.dim, device=device)
        self.ffn_norm = _param(args.dim, device=device)
        self.eps = args.norm_eps

    def forward(self, x, pending, cos, sin, varlens=None):
        if FUSE and varlens is None and _fuses_norm(x, pending, self.attn.wqkv.shape[0], two_terms_ok=self.attn.rotary_type != "llama"):
            # small decode batches: residual add + attn_norm run as the prologue of the qkv projection (round 6: the Llama
            # blocks' fusion, bit-identical, one launch less per layer)
            x, a = self.attn.decode_from_residual(x, pending, self.attn_norm, self.eps, cos, sin)
            a = tp.defer_all_reduce(a)
        else:
            x, hn = tp.add_norm(x, pending, self.attn_norm, self.eps)[:2]
            if varlens is None:
                a = tp.defer_all_reduce(self.attn.decode_forward_paged(hn, cos, sin))
            else:
                a = tp.defer_all_reduce(self.attn.prefill_forward(hn, cos, sin, varlens))
        # ffn_norm with the experts' per-token int8 quantisation of its output in the same launch (round 6: quant_act's arithmetic
        # on the same bf16 values, bit-identical codes; decode only -- prefill keeps the separate launch over thousands of rows)
        if FUSE and varlens is None and x.is_cuda:
            x, hn, hq, hs = tp.add_norm(x, a, self.ffn_norm, self.eps, quant="int8")
            # the top-2 sum moves into the next residual add whenever nothing else needs the summed tensor (one rank, or the
            # in-graph all-reduce takes the terms): one launch less per layer at every decode batch size
            defer = tp.defers_topk_sum(hn.shape[0], hn.shape[1], self.ffn.topk)
            return x, tp.defer_all_reduce(self.ffn(hn, x_quant=(hq, hs), defer_sum=defer))
        x, hn = tp.add_norm(x, a, self.ffn_norm, self.eps)[:2]
        return x, tp.defer_all_reduce(self.ffn(hn))


class MixtralDecoder(LlamaDecoder):
    """LlamaDecoder (embed, norm, head, hipGraph decode, prefill, generate) over MixtralBlocks."""

    block_type = MixtralBlock


@torch.no_grad()
def init_synthetic_(model: torch.nn.Module, seed: int = 0):
    """bf16 weights randn / sqrt(fan_in); int8 expert weights = quant_weight of such a matrix (per-channel
    scales = its row absmax / 127); norm weights 1; router weights randn / sqrt(dim)."""
    from .quantize.w8a8 import quant_weight

    dev = next(model.parameters()).device
    gen = torch.Generator(device=dev).manual_seed(seed)
    for name, p in model.named_parameters():
        if name.endswith("_scale"):
            continue
        if name.endswith("norm"):
            p.data.fill_(1.0)
        elif p.dtype == torch.int8:
            scale = dict(model.named_parameters())[name + "_scale"]
            for e in range(p.shape[0]):
                w = torch.randn(p.shape[1:], device=dev, generator=gen) * p.shape[-1] ** -0.5
                q, s = quant_weight(w)
                p.data[e].copy_(q)
                scale.data[e].copy_(s)
        else:
            std = 1.0 if name == "embed_weight" else p.shape[-1] ** -0.5
            p.data.copy_((torch.randn(p.shape, device=dev, generator=gen) * std).to(p.dtype))
    return model
