"""DeepSeek-V3/R1 decode step on the HIP operator surface (the caller of the hot path).

This is the part of chitu/models/model_deepseek_v3.py that executes per decode token, re-wired
onto chitu_amd's ops (reference file:line in each docstring): `linear_deepseek_v3:53-106`,
`AttentionDeepSeekV3._run_linear:475-536` + `decode_forward_paged:672-699` (MLA absorb-without-
precomp), `MLPDeepSeekV3:703-772`, `GateDeepSeekV3:774-842`, `MoEDeepSeekV3.forward:921-1011`
(fused path), `TransformerBlockDeepSeekV3.forward:1100-1114`, `_post_layers:1321-1325`, and the
graph capture of `Transformer.decode` (chitu/models/model.py:538-622).

Sharding is the reference's Megatron TP (chitu/models/model.py:332-370): heads and FFN / expert
width split across ranks, `wqkv_a`, the gate and the latent KV cache replicated, one all-reduce
after `wo` and one after the FFN / MoE.  Prefill, checkpoint loading, tokenizer, scheduler are out
of scope here.
"""

import math
from dataclasses import dataclass
from typing import List, Optional

import os

import torch
import torch.nn.functional as F

from . import fused_moe, graphs, ops
from . import tensor_parallel as tp
from .attn_backend import HipAttnBackend
from .cache_manager import PagedKVCacheManager

FP8 = torch.float8_e4m3fn
BLOCK = 128


@dataclass
class DeepSeekV3Args:
    """Fields of chitu/config/models/DeepSeek-R1.yaml:6-29 (defaults = DeepSeek-R1-671B)."""

    vocab_size: int = 129280
    dim: int = 7168
    inter_dim: int = 18432
    moe_inter_dim: int = 2048
    n_layers: int = 61
    n_dense_layers: int = 3
    n_heads: int = 128
    n_routed_experts: int = 256
    n_shared_experts: int = 1
    n_activated_experts: int = 8
    n_expert_groups: int = 8
    n_limited_groups: int = 4
    route_scale: float = 2.5
    score_func: str = "sigmoid"
    q_lora_rank: int = 1536
    kv_lora_rank: int = 512
    qk_nope_head_dim: int = 128
    qk_rope_head_dim: int = 64
    v_head_dim: int = 128
    rope_theta: float = 10000.0
    rope_factor: float = 40
    norm_eps: float = 1e-6
    gate_bias: Optional[bool] = None  # reference: bias iff dim == 7168 (model_deepseek_v3.py:804-808)
    # Width of the tensor-parallel sharding used for LOCAL SHAPES.  None = the live TP group size
    # (the reference's behaviour).  bench.py sets 8 to run one TP=8 rank shard per GPU even when
    # fewer than 8 ranks are live (collectives then span the live ranks only).
    shard_degree: Optional[int] = None
    # Expert parallelism (the reference's stub: `moe_world_size = 1` hard-coded at
    # model_deepseek_v3.py:870-871, `expert_map=None  # use when ep > 1` at :1004).  moe_world_size > 1:
    # rank r holds routed experts [r*E/ep, (r+1)*E/ep) at FULL width and 1/ep of the shared experts' width;
    # must equal tp_degree() (the ranks that share the replicated activations).  moe_rank None = this
    # process's TP rank (an explicit value runs one rank's shard on a lone GPU, like shard_degree).
    moe_world_size: int = 1
    moe_rank: Optional[int] = None

    def tp_degree(self):
        return self.shard_degree if self.shard_degree is not None else tp.get_tp_size()

    def has_gate_bias(self):
        return self.dim == 7168 if self.gate_bias is None else self.gate_bias


class VarLens:
    """Ragged-batch bookkeeping of a prefill call (chitu/utils.py:84-101): same fields."""

    def __init__(self, tokens, device) -> None:
        self.cpu_lens = [len(t) for t in tokens]
        self.cpu_prefix_lens = [0]
        for n in self.cpu_lens:
            self.cpu_prefix_lens.append(self.cpu_prefix_lens[-1] + n)
        self.lens = torch.tensor(self.cpu_lens, device=device, dtype=torch.int32)
        self.prefix_lens = torch.tensor(self.cpu_prefix_lens, device=device, dtype=torch.int32)
        self.max_len = max(self.cpu_lens)
        self.total_len = self.cpu_prefix_lens[-1]
        self.position_ids = torch.cat([torch.arange(n) for n in self.cpu_lens]).to(device)


def compute_softmax_scale(args: DeepSeekV3Args) -> float:
    """chitu/models/model_deepseek_v3.py:1441-1445."""
    qk_head_dim = args.qk_nope_head_dim + args.qk_rope_head_dim
    mscale = 0.1 * math.log(args.rope_factor) + 1.0
    return (qk_head_dim**-0.5) * mscale * mscale


def precompute_freqs_cis(args: DeepSeekV3Args, max_position_embeddings: int):
    """YaRN-corrected rotary table, returned as (cos, sin) fp32 [pos, rope/2].
    Restates chitu/models/model_deepseek_v3.py:1353-1438 (same constants: beta 32/1, original
    context 4096, correction only when the table is longer than that)."""
    dim, base, factor = args.qk_rope_head_dim, args.rope_theta, args.rope_factor
    freqs = 1.0 / (base ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    original = 4096
    if max_position_embeddings > original:

        def corr_dim(rot):
            return dim * math.log(original / (rot * 2 * math.pi)) / (2 * math.log(base))

        low = max(math.floor(corr_dim(32)), 0)
        high = min(math.ceil(corr_dim(1)), dim - 1)
        if low == high:
            high += 0.001
        ramp = torch.clamp((torch.arange(dim // 2, dtype=torch.float32) - low) / (high - low), 0, 1)
        smooth = 1 - ramp
        freqs = freqs / factor * (1 - smooth) + freqs * smooth
    ang = torch.outer(torch.arange(max_position_embeddings), freqs)
    cis = torch.polar(torch.ones_like(ang), ang)  # same host routine as the reference: bit-identical table
    return cis.real.contiguous(), cis.imag.contiguous()


def linear_deepseek_v3(x, weight, weight_scale=None, bias=None, x_quant=None):
    """y = x W^T (chitu/models/model_deepseek_v3.py:53-106, W8A8 branch): quantise x per 128
    values, block-scaled fp8 GEMM.  `x_quant=(q, s)` skips the quantisation when the producer
    already emitted it (fused RMSNorm)."""
    if weight.element_size() > 1:
        return F.linear(x, weight, bias)
    assert weight_scale is not None
    if x_quant is not None and isinstance(x_quant[0], ops.TiledQuant):
        # the fused step's tile-major activations (ops.TiledQuant): same GEMM, coalesced activation loads
        y = ops.fp8_gemm_deepseek_v3(x_quant[0], None, weight, weight_scale, out_dtype=torch.bfloat16)
        if bias is not None:
            y += bias
        return y
    if x_quant is None:
        shape = x.shape
        xq, xs = ops.act_quant_deepseek_v3(x.reshape(-1, shape[-1]).contiguous(), BLOCK)
    else:
        xq, xs = x_quant
        shape = xq.shape
        xq, xs = xq.reshape(-1, shape[-1]), xs.reshape(-1, xs.shape[-1])
    y = ops.fp8_gemm_deepseek_v3(xq, xs, weight, weight_scale, out_dtype=torch.bfloat16)
    if bias is not None:
        y += bias
    return y.view(*shape[:-1], y.shape[-1])


class Fp8Linear(torch.nn.Module):
    """Weight [out, in] e4m3fn + scale [ceil(out/128), ceil(in/128)] f32
    (LinearDeepSeekV3 and its parallel variants, model_deepseek_v3.py:108-392)."""

    def __init__(self, in_features, out_features, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = torch.nn.Parameter(torch.empty(out_features, in_features, dtype=FP8, device=device), requires_grad=False)
        self.scale = torch.nn.Parameter(
            torch.empty((out_features + BLOCK - 1) // BLOCK, (in_features + BLOCK - 1) // BLOCK,
                        dtype=torch.float32, device=device), requires_grad=False)

    def forward(self, x, x_quant=None):
        return linear_deepseek_v3(x, self.weight, self.scale, x_quant=x_quant)


class RMSNormW(torch.nn.Module):
    def __init__(self, dim, eps, device=None):
        super().__init__()
        self.eps = eps
        self.weight = torch.nn.Parameter(torch.ones(dim, dtype=torch.bfloat16, device=device), requires_grad=False)


# q_norm + act_quant + wq_b GEMM + KV append as one launch (ops.mla_q_proj); CHITU_Q_PROJ_FUSED=0 keeps the two
# launches apart (A/B timing, tests of the unfused pair).
FUSE_Q_PROJ = os.environ.get("CHITU_Q_PROJ_FUSED", "1") != "0"
# split merge + W_UV projection + act_quant INSIDE the MLA decode launch (round 6, chitu_hip_mla_decode_merge_uv_quant_fp8): same
# bits, one launch less -- and slower on every box it was timed on (bs 16: 24.4 us against 11.7 + 6.8; bs 1: 15.7 against
# 7.4 + 7.9; profiles/r06_ab_mla_fused_tail.txt): the in-launch hand-off (drain + flag + poll) costs more than the kernel
# boundary it removes.  Off; kept as the measured experiment and a third merge form in the bit-identity tests.
_MLA_FUSED_TAIL = os.environ.get("CHITU_MLA_FUSED_TAIL", "0") == "1"
# ... up to one 16-token tile: with two tiles every workgroup redoes the norm + quant of 32 rows and the kernel is out of
# registers (bench bs 32: 14.7 ms/step with the two launches, 15.1 with the fused one)
_Q_PROJ_MAX_BS = 16


def _wqkv_a_splits(bs: int, n: int, k: int) -> int:
    """Cross-workgroup K split of the wqkv_a GEMM (its 2112 rows are 132 MFMA tiles: half the chip), the fp32
    halves summed by the kernel that reads them.  Built, parity-tested and measured NEUTRAL on the R1 step (same-box
    A/B, profiles/r02_ab_tail_kernels.txt: 264 workgroups last as long as 132 -- the launch is bound by its chain
    of dependent round trips, not by CU count), so it is off unless CHITU_WQKV_SPLIT=1 asks for it."""
    if os.environ.get("CHITU_WQKV_SPLIT", "0") != "1" or bs > 32 or k < 4096:
        return 1
    tiles = (n + 15) // 16
    return 2 if tiles <= 160 else 1


class AttentionDeepSeekV3(torch.nn.Module):
    """MLA, absorb-without-precomp, paged decode (model_deepseek_v3.py:394-703)."""

    def __init__(self, args: DeepSeekV3Args, layer_id, cache, attn_backend, device=None):
        super().__init__()
        tp_size = args.tp_degree()
        self.layer_id, self.cache, self.attn_backend = layer_id, cache, attn_backend
        self.dim = args.dim
        self.n_local_heads = args.n_heads // tp_size
        self.q_lora_rank, self.kv_lora_rank = args.q_lora_rank, args.kv_lora_rank
        self.qk_nope_head_dim, self.qk_rope_head_dim = args.qk_nope_head_dim, args.qk_rope_head_dim
        self.qk_head_dim = args.qk_nope_head_dim + args.qk_rope_head_dim
        self.v_head_dim = args.v_head_dim
        assert self.q_lora_rank % BLOCK == 0
        assert self.qk_nope_head_dim == BLOCK and self.v_head_dim == BLOCK, "wkv_b head halves = one 128 block each"
        H = self.n_local_heads
        if self.q_lora_rank > 0:
            self.wqkv_a = Fp8Linear(args.dim, self.q_lora_rank + self.kv_lora_rank + self.qk_rope_head_dim, device)
            self.q_norm = RMSNormW(self.q_lora_rank, args.norm_eps, device)
            self.wq_b = Fp8Linear(self.q_lora_rank, H * self.qk_head_dim, device)
        else:
            # DeepSeek-V2-Lite (BASELINE config 3): no q low-rank path.  The reference's loader already maps
            # q_proj -> wq (backend.py:460) but its attention asserts q_lora_rank > 0 (model_deepseek_v3.py:477,
            # SURVEY gap G1); here wq and wkv_a are one merged GEMM [wq (H*192) | wkv_a (576)], the reference's
            # own merge pattern for wqkv_a.
            self.wq_kv_a = Fp8Linear(args.dim, H * self.qk_head_dim + self.kv_lora_rank + self.qk_rope_head_dim, device)
        self.kv_norm = RMSNormW(self.kv_lora_rank, args.norm_eps, device)
        self.wkv_b = Fp8Linear(self.kv_lora_rank, H * (self.qk_nope_head_dim + self.v_head_dim), device)
        self.wo = Fp8Linear(H * self.v_head_dim, args.dim, device)
        self.softmax_scale = compute_softmax_scale(args)
        self._w_uk_t, self._w_uk_key = None, None

    def w_uk_transposed(self):
        """fp8 copy of W_UK laid out [H, kv_lora, nope] so the absorbed contraction index is
        contiguous.  Layout-only preprocessing of wkv_b; values and scales untouched.
        The copy is keyed on wkv_b's storage and version counter and rebuilt IN PLACE when they change
        (load_state_dict, copy_, .to()), so a captured hipGraph keeps reading the right bytes; writers that
        bypass the counter (`.data` fills: init_synthetic_, checkpoint.load_deepseek_v3) call
        `refresh_derived_layouts`."""
        wt = self.wkv_b.weight
        key = (wt.data_ptr(), None if wt.is_inference() else wt._version)  # inference tensors carry no version counter
        if self._w_uk_t is None or self._w_uk_key != key:
            H = self.n_local_heads
            w = wt.view(torch.uint8).view(H, self.qk_nope_head_dim + self.v_head_dim, self.kv_lora_rank)
            t = w[:, : self.qk_nope_head_dim].transpose(1, 2)
            if self._w_uk_t is not None and self._w_uk_t.device == wt.device:
                self._w_uk_t.view(torch.uint8).copy_(t)
            else:
                self._w_uk_t = t.contiguous().view(FP8)
            self._w_uk_key = key
        return self._w_uk_t

    def first_projection(self):
        """The attention's first linear -- wqkv_a, or [wq | wkv_a] of a model without a q low-rank path."""
        return self.wqkv_a if self.q_lora_rank > 0 else self.wq_kv_a

    def decode_forward_paged(self, x_quant, cos, sin, first=None):
        """x_quant = fp8 (q, s) of attn_norm(x), [bs, dim].  Returns wo(attn) before the all-reduce.
        first: the first projection's output [bs, N] bf16 when the launch in front already computed it
        (ops.fp8_linear_add_norm: attn_norm as that GEMM's prologue); x_quant is then not read.

        6 launches (the reference's decode_forward_paged + _run_linear issue ~25): wqkv_a GEMM,
        [q_norm + quant -> wq_b GEMM | kv_norm + RoPE(k_pe) + page append], [W_UK absorb | RoPE(q_pe)],
        MLA decode, [split merge + W_UV absorb + quant], wo GEMM.  (7 with batches above 16: the q_norm / kv
        launch and the wq_b GEMM apart.)"""
        H, C, R = self.n_local_heads, self.kv_lora_rank, self.qk_rope_head_dim
        if first is not None:
            bs = first.shape[0]
        else:
            bs = x_quant[0].rows if isinstance(x_quant[0], ops.TiledQuant) else x_quant[0].shape[0]
        cache = self.cache
        kv_cache = cache.get_paged_kv_cache(self.layer_id)
        nblk = C // BLOCK
        if self.q_lora_rank > 0:
            # wqkv_a: [bs, q_lora + C + R] (optionally as split-K planes, see _wqkv_a_splits)
            splits = _wqkv_a_splits(bs, self.wqkv_a.out_features, self.wqkv_a.in_features)
            if first is not None or isinstance(x_quant[0], ops.TiledQuant):
                splits = 1
            if first is not None:
                q_a_kv = first
            elif splits > 1:
                q_a_kv = ops.fp8_gemm_partials_deepseek_v3(x_quant[0], x_quant[1], self.wqkv_a.weight, self.wqkv_a.scale, splits)
            else:
                q_a_kv = self.wqkv_a(None, x_quant=x_quant)
            if FUSE_Q_PROJ and splits == 1 and bs <= _Q_PROJ_MAX_BS and ops.mla_q_proj_fits(bs, self.q_lora_rank):
                # ONE launch: q_norm + act_quant as the prologue of the wq_b GEMM, and this token's
                # [kv_norm(kv_c) | rope(k_pe)] row straight into its page on extra workgroups of the same grid
                q = ops.mla_q_proj(q_a_kv, self.q_lora_rank, self.q_norm.weight, self.q_norm.eps, self.wq_b.weight,
                                   self.wq_b.scale, self.kv_norm.weight, self.kv_norm.eps, cos, sin, kv_cache,
                                   cache.get_gpu_block_table(), cache.get_gpu_seq_lens_excl_this_decode(),
                                   out_dtype=torch.bfloat16)
            else:
                # q_norm + quant, and this token's [kv_norm(kv_c) | rope(k_pe)] row straight into its page
                qq, qs = ops.mla_qkv_post(q_a_kv, self.q_lora_rank, self.q_norm.weight, self.q_norm.eps, self.kv_norm.weight,
                                          self.kv_norm.eps, cos, sin, kv_cache, cache.get_gpu_block_table(),
                                          cache.get_gpu_seq_lens_excl_this_decode())
                q = self.wq_b(None, x_quant=(qq, qs))
            q = q.view(bs, H, self.qk_head_dim)
            q_nope, q_pe = q[..., : self.qk_nope_head_dim], q[..., self.qk_nope_head_dim :]
            # q_nope' = q_nope . W_UK  (einsum "shd,hdc->shc", :529-531), wkv_b dequantised in registers;
            # q_pe rotated in place by the same launch
            q_abs = ops.absorb_bmm_rope_fp8(q_nope, self.w_uk_transposed(), self.wkv_b.scale, 0, 2 * nblk, 1, 0, q_pe, cos, sin)
        else:
            q_kv = first if first is not None else self.wq_kv_a(None, x_quant=x_quant)  # [bs, H*192 + C + R]
            nq = H * self.qk_head_dim
            q = q_kv[:, :nq].view(bs, H, self.qk_head_dim)
            q_nope, q_pe = q[..., : self.qk_nope_head_dim], q[..., self.qk_nope_head_dim :]
            # ONE launch: W_UK absorb, RoPE(q_pe) in place, and this token's [kv_norm(kv_c) | rope(k_pe)] row into its page
            q_abs = ops.absorb_bmm_rope_kv_fp8(q_nope, self.w_uk_transposed(), self.wkv_b.scale, 0, 2 * nblk, 1, 0, q_pe, cos, sin,
                                               q_kv[:, nq:], self.kv_norm.weight, self.kv_norm.eps, kv_cache,
                                               cache.get_gpu_block_table(), cache.get_gpu_seq_lens_excl_this_decode())
        # small batches: the split-KV merge runs inside the W_UV projection kernel (CHITU_MLA_FUSED_TAIL=1: both inside the
        # decode launch, same bits, measured slower -- see _MLA_FUSED_TAIL)
        fuse_merge = bs <= 32 and C == 512
        w_uv = self.wkv_b.weight.view(H, self.qk_nope_head_dim + self.v_head_dim, C)[:, self.qk_nope_head_dim :]
        fused = None
        if fuse_merge and _MLA_FUSED_TAIL and self.v_head_dim == 128:
            fused = self.attn_backend.mla_decode_merge_uv_quant(
                q_abs, q_pe, kv_cache, cache.get_gpu_seq_lens_incl_this_decode(), cache.get_gpu_block_table(), self.softmax_scale,
                w_uv, self.wkv_b.scale, nblk, 2 * nblk, 1, tile_major=ops.tile_major_ok(bs))
        if fused is not None:
            oq, os_ = fused
            return self.wo(None, x_quant=(oq, os_))
        o = self.attn_backend.mla_decode(q_abs, q_pe, kv_cache, cache.get_gpu_seq_lens_incl_this_decode(),
                                         cache.get_gpu_block_table(), self.softmax_scale, return_partials=fuse_merge)
        # out = o . W_UV^T  (einsum "bshc,hdc->bshd", :697) + the act-quant of wo's input
        if isinstance(o, tuple):
            oq, os_ = ops.mla_merge_absorb_uv_quant_fp8(o[0], o[1], bs, w_uv, self.wkv_b.scale, nblk, 2 * nblk, 1,
                                                        tile_major=ops.tile_major_ok(bs))
        else:
            oq, os_ = ops.absorb_uv_quant_fp8(o, w_uv, self.wkv_b.scale, nblk, 2 * nblk, 1)
        return self.wo(None, x_quant=(oq, os_))


    def prefill_forward(self, x_quant, cos, sin, varlens):
        """Prefill in absorb mode (model_deepseek_v3.py:538-603): the same projections as decode on all
        T prompt tokens, [kv_norm(kv_c) | rope(k_pe)] rows written to the pages by the cache manager
        (cache_manager.py:93-142), causal MQA through attn_backend.attn_varlen_func.  Op-level launches
        (prefill is outside the decode hot path; SURVEY 8f.1 first cut)."""
        H, C, R = self.n_local_heads, self.kv_lora_rank, self.qk_rope_head_dim
        T = x_quant[0].shape[0]
        if self.q_lora_rank > 0:
            q_a_kv = self.wqkv_a(None, x_quant=x_quant)  # [T, q_lora + C + R]
            ql = self.q_lora_rank
            _, qq, qs = ops.rms_norm(q_a_kv[:, :ql], self.q_norm.weight, self.q_norm.eps, out_bf16=False, quant="act")
            q = self.wq_b(None, x_quant=(qq, qs)).view(T, H, self.qk_head_dim)
        else:
            q_a_kv = self.wq_kv_a(None, x_quant=x_quant)  # [T, H*192 + C + R]
            ql = H * self.qk_head_dim
            q = q_a_kv[:, :ql].view(T, H, self.qk_head_dim)
        q_nope, q_pe = q[..., : self.qk_nope_head_dim], q[..., self.qk_nope_head_dim :]
        kv_c = ops.rms_norm(q_a_kv[:, ql : ql + C], self.kv_norm.weight, self.kv_norm.eps)
        q_pe, k_pe = ops.apply_rotary_pos_emb(q_pe, q_a_kv[:, ql + C :], cos, sin, rotary_type="llama")
        kv_pe = torch.cat([kv_c, k_pe], dim=-1)  # [T, C + R]
        self.cache.finalize_cache_bylayer_prefill(kv_pe, None, self.cache.curr_req_ids, self.cache.curr_varlens, self.layer_id)
        nblk = C // BLOCK
        q_abs = ops.absorb_bmm_fp8(q_nope, self.w_uk_transposed(), self.wkv_b.scale, 0, 2 * nblk, 1, 0)
        o = self.attn_backend.attn_varlen_func(
            torch.cat([q_abs, q_pe], dim=-1), kv_pe.view(T, 1, C + R), kv_c.view(T, 1, C), varlens.prefix_lens,
            varlens.prefix_lens, varlens.max_len, varlens.max_len, causal=True, softmax_scale=self.softmax_scale)
        w_uv = self.wkv_b.weight.view(H, self.qk_nope_head_dim + self.v_head_dim, C)[:, self.qk_nope_head_dim :]
        oq, os_ = ops.absorb_uv_quant_fp8(o, w_uv, self.wkv_b.scale, nblk, 2 * nblk, 1)
        return self.wo(None, x_quant=(oq, os_))


class MLPDeepSeekV3(torch.nn.Module):
    """Dense FFN of the first n_dense_layers (model_deepseek_v3.py:703-772), gate/up merged."""

    def __init__(self, args, device=None, inter: Optional[int] = None):
        super().__init__()
        tp_size = args.tp_degree()
        self.inter = args.inter_dim // tp_size if inter is None else inter
        self.w1w3 = Fp8Linear(args.dim, 2 * self.inter, device)
        self.w2 = Fp8Linear(self.inter, args.dim, device)

    def forward(self, x_quant):
        h = self.w1w3(None, x_quant=x_quant)
        hq, hs = fused_moe.silu_and_mul_quant(h, mode="act")
        return self.w2(None, x_quant=(hq, hs))


_ROUTE_ALIGN_MAX_TOKENS = 64


def _route_align_enabled(tokens: int) -> bool:
    """Routing + moe_align in one launch for decode-sized batches (the sort is done by the last routing
    workgroup: fine for a few hundred ids, slower than the 1024-thread align kernel for prefill-sized ones)."""
    return tokens <= _ROUTE_ALIGN_MAX_TOKENS and os.environ.get("CHITU_ROUTE_ALIGN", "1") != "0"


class GateDeepSeekV3(torch.nn.Module):
    """Router (model_deepseek_v3.py:774-842): bf16 scores, sigmoid/softmax, bias, group-limited
    top-k, normalise, route_scale."""

    def __init__(self, args, device=None):
        super().__init__()
        self.topk, self.n_groups, self.topk_groups = args.n_activated_experts, args.n_expert_groups, args.n_limited_groups
        self.score_func, self.route_scale = args.score_func, args.route_scale
        self.weight = torch.nn.Parameter(torch.empty(args.n_routed_experts, args.dim, dtype=torch.bfloat16, device=device), requires_grad=False)
        self.bias = (torch.nn.Parameter(torch.empty(args.n_routed_experts, dtype=torch.bfloat16, device=device), requires_grad=False)
                     if args.has_gate_bias() else None)

    def forward(self, x, extra_expert_id: int = -1, extra_count: int = 1, align=None, logits_partials=None):
        """(weights [bs, topk(+extra)] bf16, indices int64).  Two HIP launches (ops.gate_deepseek_v3).
        align=(num_experts, block, expert_map): also the moe_align triple, sorted inside the routing launch.
        logits_partials: the score GEMM's split-K planes when a launch in front already produced them; x is then not read."""
        return ops.gate_deepseek_v3(x, self.weight, self.bias, self.n_groups, self.topk_groups, self.topk,
                                    self.score_func, self.route_scale, extra_expert_id=extra_expert_id,
                                    extra_count=extra_count, align=align, logits_partials=logits_partials)


class MoEDeepSeekV3(torch.nn.Module):
    """Routed + shared experts, fused path (model_deepseek_v3.py:845-1011).  Experts are stacked
    [n_routed + n_shared, ...] with the shared expert last, every rank holds all experts at 1/tp
    width (:883-919)."""

    def __init__(self, args, device=None):
        super().__init__()
        tp_size = args.tp_degree()
        self.n_routed, self.n_shared = args.n_routed_experts, args.n_shared_experts
        self.moe_world_size = args.moe_world_size
        self.gate = GateDeepSeekV3(args, device)
        if self.moe_world_size > 1:
            # expert parallel: names follow model_deepseek_v3.py:870-880
            ep = self.moe_world_size
            assert tp_size == ep, "experts are partitioned over the ranks that share the activations (tp group)"
            assert self.n_routed % ep == 0, f"Number of experts must be divisible by world size (world_size={ep})"
            self.moe_rank = (tp.get_tp_rank() % ep) if args.moe_rank is None else args.moe_rank
            self.n_local_experts = self.n_routed // ep
            self.experts_start_idx = self.moe_rank * self.n_local_experts
            self.experts_end_idx = self.experts_start_idx + self.n_local_experts
            emap = torch.full((self.n_routed,), -1, dtype=torch.int32)
            emap[self.experts_start_idx:self.experts_end_idx] = torch.arange(self.n_local_experts, dtype=torch.int32)
            self.register_buffer("expert_map", emap.to(device) if device is not None else emap, persistent=False)
            self.inter = args.moe_inter_dim
            E = self.n_local_experts
            shared_inter = self.n_shared * args.moe_inter_dim
            assert shared_inter % (ep * BLOCK) == 0 or self.n_shared == 0, "shared width / ep must keep 128-blocks whole"
            # n_shared experts of width I on the same input = one MLP of width n_shared * I; its width is
            # split over the ranks like any row/column-parallel MLP and rides in the layer's all-reduce
            self.shared = MLPDeepSeekV3(args, device, inter=shared_inter // ep) if self.n_shared else None
        else:
            self.inter = args.moe_inter_dim // tp_size
            E = self.n_routed + self.n_shared
        self.w1w3_weight = torch.nn.Parameter(torch.empty(E, 2 * self.inter, args.dim, dtype=FP8, device=device), requires_grad=False)
        self.w1w3_scale = torch.nn.Parameter(torch.empty(E, (2 * self.inter + BLOCK - 1) // BLOCK, args.dim // BLOCK, dtype=torch.float32, device=device), requires_grad=False)
        self.w2_weight = torch.nn.Parameter(torch.empty(E, args.dim, self.inter, dtype=FP8, device=device), requires_grad=False)
        self.w2_scale = torch.nn.Parameter(torch.empty(E, args.dim // BLOCK, (self.inter + BLOCK - 1) // BLOCK, dtype=torch.float32, device=device), requires_grad=False)

    def forward(self, x, x_quant, defer_sum: bool = False, logits_partials=None):
        """x: ffn_norm output bf16 [bs, dim] (gate input); x_quant its fp8 (per-group) form.
        logits_partials: the router's score planes if ffn_norm's launch computed them (then x is only the output buffer).
        defer_sum: return the un-summed [bs, topk + n_shared, dim] expert outputs; the next RMSNorm folds
        the top-k sum into its residual add (ops.rms_norm(add=<3-D>)).

        The shared experts (1 in V3/R1, 2 in V2-Lite) are routed as slots topk .. topk+n_shared-1 with
        weight 1 and run inside the same grouped GEMMs as the routed experts: their tokens form full MFMA
        tiles and four launches per shared expert disappear.  Differences to the reference (:936-949,
        1010), both inside its tolerance: shared + routed are summed in fp32 and rounded once instead of
        bf16 + bf16, and the shared experts' input uses the group-quant rule of the routed ones."""
        nr, ns = self.n_routed, self.n_shared
        if self.moe_world_size > 1:
            return self.forward_expert_parallel(x, x_quant)
        # decode-sized batches: moe_align runs inside the routing launch (last workgroup sorts)
        align = (nr + ns, fused_moe._MOE_BLOCK_M, None) if _route_align_enabled(x.shape[0]) else None
        routed = (self.gate(x, extra_expert_id=nr, extra_count=ns, align=align, logits_partials=logits_partials) if ns >= 1
                  else self.gate(x, align=align, logits_partials=logits_partials))
        weights, indices = routed[0], routed[1]
        aligned = routed[2] if len(routed) > 2 else None
        return fused_moe.fused_experts(
            x, self.w1w3_weight, self.w2_weight, topk_weights=weights, topk_ids=indices, use_fp8_w8a8=True,
            inplace=True, global_num_experts=nr + ns, w1_scale=self.w1w3_scale, w2_scale=self.w2_scale,
            block_shape=[BLOCK, BLOCK], a1_quant=x_quant, reduce_topk=not defer_sum, aligned=aligned,
        )

    def forward_expert_parallel(self, x, x_quant):
        """Expert-parallel MoE.  The activations are replicated over the group (the attention
        all-reduce leaves them so), every rank routes ALL tokens with the replicated gate, runs the slots
        whose expert it holds (expert_map: global id -> local id, -1 = another rank's, which the grouped
        GEMMs zero-fill like write_zeros_to_output, fused_moe.py:40-59) at full expert width, adds its slice
        of the shared experts, and the layer's existing all-reduce is the combine -- no all-to-all is
        needed while the tokens are replicated.  Returns this rank's partial sum [bs, dim]."""
        align = (self.n_routed, fused_moe._MOE_BLOCK_M, self.expert_map) if _route_align_enabled(x.shape[0]) else None
        routed = self.gate(x, align=align)
        weights, indices = routed[0], routed[1]
        y = fused_moe.fused_experts(
            x, self.w1w3_weight, self.w2_weight, topk_weights=weights, topk_ids=indices, use_fp8_w8a8=True,
            inplace=True, global_num_experts=self.n_routed, expert_map=self.expert_map, w1_scale=self.w1w3_scale,
            w2_scale=self.w2_scale, block_shape=[BLOCK, BLOCK], a1_quant=x_quant,
            aligned=routed[2] if len(routed) > 2 else None,
        )
        if self.shared is not None:
            y += self.shared(x_quant)
        return y


class TransformerBlockDeepSeekV3(torch.nn.Module):
    """x += attn(attn_norm(x)); x += ffn(ffn_norm(x))  (model_deepseek_v3.py:1064-1114)."""

    def __init__(self, layer_id, args, cache, attn_backend, device=None):
        super().__init__()
        self.attn = AttentionDeepSeekV3(args, layer_id, cache, attn_backend, device)
        self.is_moe = layer_id >= args.n_dense_layers
        self.ffn = MoEDeepSeekV3(args, device) if self.is_moe else MLPDeepSeekV3(args, device)
        self.attn_norm = RMSNormW(args.dim, args.norm_eps, device)
        self.ffn_norm = RMSNormW(args.dim, args.norm_eps, device)

    def forward(self, x, pending, cos, sin, varlens=None):
        """(x, pending) -> (x', pending'); varlens given = prefill (ragged prompt tokens), else decode.
        The residual stream is x + pending; every residual add is folded into the RMSNorm that consumes the sum
        (`add_norm`), so a layer is norm, attention, norm, ffn with no separate add launches (reference:
        :1107-1113).  With tensor parallelism the sublayer outputs are partials: `tp.defer_all_reduce` hands
        their all-reduce to that same norm launch when the in-graph xGMI collectives are on (then a layer has
        no stand-alone collective at all), and performs it through the library otherwise."""
        # decode: the fp8 form of the normed row goes to wqkv_a only -> tile-major (ops.TiledQuant), wherever a residual
        # is folded in (every layer but the first)
        tm = varlens is None and pending is not None and ops.tile_major_ok(x.shape[0])
        if varlens is None and _fuses_attn_norm_into_first_projection(x, pending, self.attn):
            # batch 1: [the experts' top-k sum +] residual add + attn_norm + act_quant run as the prologue of the wqkv_a
            # GEMM, redone by each of its workgroups -- one launch less per layer, bit-identical
            proj = self.attn.first_projection()
            x, first = ops.fp8_linear_add_norm(x, pending, self.attn_norm.weight, self.attn_norm.eps, proj.weight, proj.scale)
            a = tp.defer_all_reduce(self.attn.decode_forward_paged(None, cos, sin, first=first))
            return self.ffn_part(x, a, cos, sin, varlens)
        x, _, xq, xs = add_norm(x, pending, self.attn_norm, out_bf16=False, quant="act", tile_major=tm)
        if varlens is None:
            a = tp.defer_all_reduce(self.attn.decode_forward_paged((xq, xs), cos, sin))
        else:
            a = tp.defer_all_reduce(self.attn.prefill_forward((xq, xs), cos, sin, varlens))
        return self.ffn_part(x, a, cos, sin, varlens)

    def ffn_part(self, x, a, cos, sin, varlens):
        """The layer's second half: ffn_norm (+ the residual add of the attention output) and the MLP / MoE."""
        if self.is_moe:
            x, hn, hq, hs = add_norm(x, a, self.ffn_norm, out_bf16=True, quant="group")
            # the experts' top-k sum moves into the next norm launch whenever nothing else needs the summed
            # tensor: always on one rank, and under TP when the all-reduce is that launch too
            defer = (tp.defers_topk_sum(hn.shape[0], hn.shape[1], self.ffn.gate.topk + self.ffn.n_shared)
                     and self.ffn.moe_world_size == 1 and os.environ.get("CHITU_DEFER_TOPK_SUM", "1") != "0")
            f = self.ffn(hn, (hq, hs), defer_sum=defer)
        else:
            x, _, hq, hs = add_norm(x, a, self.ffn_norm, out_bf16=False, quant="act",
                                    tile_major=varlens is None and ops.tile_major_ok(x.shape[0]))
            f = self.ffn((hq, hs))
        return x, tp.defer_all_reduce(f)


# Decode batches up to this size run attn_norm as the prologue of the attention's first projection
# (ops.fp8_linear_add_norm); 0 = off.  Same-box A/B on the R1 step at bs 1: 4.587 -> 4.451 ms/step (the fused launch
# lasts 10.2 us against 5.9 + 7.1 for the pair; profiles/r04_ab_norm_prologues.txt); a two-row batch only qualifies
# behind a plain residual (no top-k terms) and is neutral.
FUSE_ATTN_NORM_MAX_BS = int(os.environ.get("CHITU_FUSE_ATTN_NORM_MAX_BS", "1"))


def _fuses_attn_norm_into_first_projection(x, pending, attn) -> bool:
    """pending: a plain [bs, dim] tensor or the experts' un-summed [bs, terms, dim] (not None: the first layer; not a partial
    whose all-reduce the norm launch itself performs); a shape the fused launch takes; the split-K form of wqkv_a off."""
    if not (isinstance(pending, torch.Tensor) and pending.dim() in (2, 3) and x.shape[0] <= FUSE_ATTN_NORM_MAX_BS):
        return False
    proj = attn.first_projection()
    terms = pending.shape[1] if pending.dim() == 3 else 1
    return (proj.weight.element_size() == 1 and os.environ.get("CHITU_WQKV_SPLIT", "0") != "1"
            and ops.fp8_linear_add_norm_fits(x.shape[0], proj.out_features, proj.in_features, terms))


# (ffn_norm as the prologue of the router's score GEMM was built in round 4, bit-identical, measured slower on the R1 step --
# bs 1 4.587 -> 4.668 ms, profiles/r04_ab_norm_prologues.txt: the score GEMM's 256 workgroups stream 14 KB of weights each and
# would each redo a 43 KB norm -- and removed in round 5.)


def add_norm(x, pending, norm, out_bf16=True, quant=None, tile_major=False):
    """tensor_parallel.add_norm on an RMSNormW module: residual add (+ the all-reduce of a deferred partial) + norm
    [+ fp8 quant] in one launch; returns (x_new, y, q, s) (q an ops.TiledQuant and s None under tile_major)."""
    return tp.add_norm(x, pending, norm.weight, norm.eps, out_bf16=out_bf16, quant=quant, tile_major=tile_major)


class DeepSeekV3Decoder(torch.nn.Module):
    """Decode-only transformer: embed -> blocks -> norm -> head -> fp32 logits, with the whole
    step captured as a hipGraph per batch size (chitu/models/model.py:538-622; no third-party
    attention gate as at :543-546 -- the HIP backend is capture-safe by construction)."""

    def __init__(self, args: DeepSeekV3Args, cache: PagedKVCacheManager, attn_backend: HipAttnBackend,
                 max_position_embeddings: int = 4096, device="cuda", layers: Optional[List[int]] = None):
        super().__init__()
        self.args, self.cache, self.attn_backend, self.device = args, cache, attn_backend, torch.device(device)
        deg = args.tp_degree()
        assert args.vocab_size % deg == 0
        self.vocab_local = args.vocab_size // deg
        self.vocab_start = (tp.get_tp_rank() % deg) * self.vocab_local
        # VocabParallelEmbedding / ColumnParallelLinear(gather_output) of the reference
        # (model_deepseek_v3.py:1292-1319), sharded `deg` ways
        self.embed_weight = torch.nn.Parameter(torch.empty(self.vocab_local, args.dim, dtype=torch.bfloat16, device=device), requires_grad=False)
        ids = list(range(args.n_layers)) if layers is None else layers
        self.layers = torch.nn.ModuleList(TransformerBlockDeepSeekV3(i, args, cache, attn_backend, device) for i in ids)
        self.norm = RMSNormW(args.dim, args.norm_eps, device)
        self.head_weight = torch.nn.Parameter(torch.empty(self.vocab_local, args.dim, dtype=torch.bfloat16, device=device), requires_grad=False)
        cos, sin = precompute_freqs_cis(args, max_position_embeddings)
        self.cos_table, self.sin_table = cos.to(device), sin.to(device)
        self.graphs, self.static_tokens, self.static_out = {}, {}, {}
        self.graph_pool = None

    def decode_eager(self, tokens):
        """tokens [bs] int64 -> logits [bs, vocab] fp32 (decode_single_device, model.py:468-475)."""
        # embedding rows of this rank's vocabulary slice + every sequence's rotary row (prepare_freqs_cis_decode,
        # model.py:429-448): one launch
        h, cos, sin = ops.embed_rope_gather(tokens, self.embed_weight, self.vocab_start if self.vocab_local != self.args.vocab_size else 0,
                                            self.cache.get_gpu_seq_lens_excl_this_decode(), self.cos_table, self.sin_table)
        if self.vocab_local != self.args.vocab_size:
            h = tp.all_reduce(h)  # tensor_parallel.py:199-208
        pending = None
        for layer in self.layers:
            h, pending = layer(h, pending, cos, sin)
        h = add_norm(h, pending, self.norm)[1]
        # bf16 logits -> fp32 (model.py:475), the cast riding in the gather
        return tp.all_gather_last_dim(ops.bf16_linear(h, self.head_weight), out_dtype=torch.float32)

    @torch.inference_mode()
    def prefill(self, tokens, req_ids):
        """tokens: list of per-request token-id lists; req_ids: their cache keys.  Runs the prompt
        through every layer, fills the KV pages, returns fp32 logits [n_req, vocab] of each prompt's
        LAST token (prefill_single_device, model.py:451-465).  Eager launches."""
        from . import graphs

        graphs.sweep_if_memory_was_recycled()  # the eager path's canary: once after an xGMI communicator came or went
        varlens = VarLens(tokens, self.device)
        self.cache.curr_varlens, self.cache.curr_req_ids = varlens, list(req_ids)
        flat = torch.tensor([t for seq in tokens for t in seq], dtype=torch.int64, device=self.device)
        cos, sin = self.cos_table[varlens.position_ids], self.sin_table[varlens.position_ids]
        h, pending = self.embed(flat), None
        for layer in self.layers:
            h, pending = layer(h, pending, cos, sin, varlens)
        last = torch.tensor([p - 1 for p in varlens.cpu_prefix_lens[1:]], dtype=torch.int64, device=self.device)
        h = h[last]
        pending = tp.resolve(pending)
        if pending is not None:
            pending = pending[last].contiguous()
        h = add_norm(h, pending, self.norm)[1]
        self.cache.finalize_cache_all_prefill(req_ids, varlens)
        return tp.all_gather_last_dim(ops.bf16_linear(h, self.head_weight), out_dtype=torch.float32)

    @torch.inference_mode()
    def generate(self, prompts, max_new_tokens, req_ids=None, use_graph=True, temperatures=None, top_ks=None,
                 top_ps=None, frequency_penalties=None, generator=None):
        """Prefill, then max_new_tokens - 1 decode steps; returns [n_req, max_new_tokens] int64 and frees the
        requests' pages.  Token selection is executor.py:82-112 on the device (chitu_amd.sampling): greedy
        unless some top_k > 1 (then per-request temperature / top-k / top-p sampling, uniforms from
        `generator`), with an optional per-request frequency penalty; tokens never visit the host."""
        from .sampling import DeviceSampler

        req_ids = [f"gen{i}" for i in range(len(prompts))] if req_ids is None else list(req_ids)
        pick = DeviceSampler(len(prompts), max_new_tokens, self.device, temperatures, top_ks, top_ps,
                             frequency_penalties, generator)
        tok = pick(self.prefill(prompts, req_ids))
        out = [tok]
        for _ in range(max_new_tokens - 1):
            self.cache.prepare_cache_decode(req_ids)
            self.cache.prepare_block_table_for_decode(req_ids)
            tok = pick(self.decode(tok, use_graph=use_graph))
            self.cache.finalize_cache_single_decode(req_ids)
            out.append(tok)
        for r in req_ids:
            self.cache.finalize_cache_all_decode(r)
        tokens_out = torch.stack(out, dim=1)
        if tp.xgmi_comm() is not None:
            # the tokens are about to leave the engine: make sure no collective behind them gave up on a peer
            torch.cuda.current_stream().synchronize()
            tp.check_comm()
        return tokens_out

    def embed(self, tokens):
        """tensor_parallel.py:199-208: mask ids outside this rank's vocab slice, lookup, all-reduce."""
        if self.vocab_local == self.args.vocab_size:
            return F.embedding(tokens, self.embed_weight)
        local = tokens - self.vocab_start
        mask = (local < 0) | (local >= self.vocab_local)
        y = F.embedding(torch.where(mask, torch.zeros_like(local), local), self.embed_weight)
        y = torch.where(mask.unsqueeze(-1), torch.zeros_like(y), y)
        return tp.all_reduce(y)

    def prepare_decoding_attn(self):
        """model_deepseek_v3.py:1339-1350 -- outside the graph."""
        c = self.cache
        self.attn_backend.prepare_metadata_for_decode(
            c.get_gpu_seq_lens_excl_this_decode(), c.get_gpu_seq_lens_incl_this_decode(),
            c.get_gpu_block_table(), c.get_block_size(), softmax_scale=compute_softmax_scale(self.args))

    @torch.inference_mode()
    def decode(self, tokens, use_graph=True):
        """One decode step -> fp32 logits [bs, vocab].  use_graph: False = eager launches; True = replay of
        the step captured per batch size -- as ONE hipGraph when this rank has no collectives (or when
        CHITU_TP_GRAPH=full asks for the collectives to be captured too), else PIECEWISE: the step is cut at
        every all-reduce / all-gather, the pieces are hipGraphs and the collectives are issued between them
        (RCCL never has to be capturable; 2 x layers + 2 host calls per step, far below the pieces' GPU
        time); "piecewise" / "full" force a mode."""
        tp.check_comm()  # a collective of an earlier step that timed out: raise instead of decoding garbage
        self.prepare_decoding_attn()
        bs = tokens.shape[0]
        if not use_graph or tp.xgmi_split_phase():  # (split-phase collectives hold a host barrier: not capturable)
            return self.decode_eager(tokens)
        mode = graphs.graph_mode(use_graph)
        key = (bs, mode)
        if bs not in self.static_tokens:
            self.static_tokens[bs] = tokens.clone()
        else:
            self.static_tokens[bs].copy_(tokens)
        if key not in self.graphs:
            # Eager step on the static inputs (it replicates the graph's side effect, the append at position L, which the
            # replays then overwrite with the same values), capture, ONE replay checked bit for bit against that eager
            # step (graphs.capture_verified): a graph that does not reproduce the eager launches is never used.
            g, self.graph_pool, self.static_out[bs] = graphs.capture_verified(
                lambda: self.decode_eager(self.static_tokens[bs]), self.static_out.get(bs), mode, self.graph_pool,
                what=f"DeepSeekV3Decoder decode step bs={bs}")
            self.graphs[key] = g
        self.graphs[key].replay()
        return self.static_out[bs]


def refresh_derived_layouts(model: torch.nn.Module):
    """Rebuild every layout derived from the weights (the transposed W_UK copies) after the weights were written
    through a path the version counters do not see."""
    for mod in model.modules():
        if isinstance(mod, AttentionDeepSeekV3) and mod._w_uk_t is not None:
            mod._w_uk_key = None
            mod.w_uk_transposed()


# ---------------------------------------------------------------- synthetic weights
def _fill_fp8(t: torch.Tensor, gen: torch.Generator, std=0.5, chunk=1 << 26):
    flat = t.view(-1)
    for i in range(0, flat.numel(), chunk):
        n = min(chunk, flat.numel() - i)
        flat[i : i + n].copy_((torch.randn(n, device=t.device, dtype=torch.bfloat16, generator=gen) * std).to(FP8))


@torch.no_grad()
def init_synthetic_(model: torch.nn.Module, seed: int = 0, router_std: float = None):
    """fp8 = (randn*0.5) -> e4m3 (never raw bytes: no NaN codes), block scales U(0.01,0.03),
    bf16 weights randn*0.05, norm weights 1, gate bias small, router weights randn/sqrt(dim)
    (unit-variance logits => near-uniform routing) -- SURVEY.md 8(d)."""
    dev = next(model.parameters()).device
    gen = torch.Generator(device=dev).manual_seed(seed)
    for name, p in model.named_parameters():
        if p.dtype == FP8:
            _fill_fp8(p.data, gen)
        elif p.dtype == torch.float32:
            p.data.copy_(torch.rand(p.shape, device=dev, generator=gen) * 0.02 + 0.01)
        elif name.endswith("norm.weight"):
            p.data.fill_(1.0)
        elif name.endswith("gate.bias"):
            p.data.copy_((torch.randn(p.shape, device=dev, generator=gen) * 0.01).to(p.dtype))
        elif name.endswith("gate.weight"):
            # router logits ~ N(0, 1): with the generic 0.05 the sigmoid scores saturate, the top-k is
            # decided by the (token-independent) bias and every token picks the same experts, which
            # under-counts the expert bytes a balanced router (R1's is) streams per step
            std = p.shape[-1] ** -0.5 if router_std is None else router_std
            p.data.copy_((torch.randn(p.shape, device=dev, generator=gen) * std).to(p.dtype))
        else:
            flat = p.data.view(-1)
            for i in range(0, flat.numel(), 1 << 26):
                n = min(1 << 26, flat.numel() - i)
                flat[i : i + n].copy_((torch.randn(n, device=dev, generator=gen) * 0.05).to(p.dtype))
    refresh_derived_layouts(model)
    return model
