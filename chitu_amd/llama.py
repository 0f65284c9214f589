"""Llama-family (GQA / MHA, bf16) decode step on the gfx950 kernels -- BASELINE config 2
("Llama-3-8B bf16 TP=1 on one MI355X, paged KV, hipGraph decode").

Reference (read-only): chitu/models/model.py -- Attention:81-198 (decode_forward_paged:167-198),
FeedForward:201-214, TransformerBlock, Transformer.decode:538-622; models/model_llama.py (merged
wqkv / w13 layout of the original Meta checkpoints, rotary_type "llama" = interleaved pairs).

Per layer and step, 7 launches: add+RMSNorm, wqkv GEMM, [RoPE(q, k) + append K, V], paged GQA decode
(+ merge when the KV range is split), wo GEMM, add+RMSNorm, [w13 GEMM + SiluAndMul], w2 GEMM -- 4 at batch 1-2,
where both add+RMSNorm steps run as the prologue of the GEMM behind them and RoPE + the K / V append as the
epilogue of the qkv projection (bf16_norm_gemm.hip).
All GEMMs are the weight-streaming skinny bf16 kernel (gate.hip); the whole step replays as one hipGraph.
"""

import os
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from . import graphs, ops
from . import tensor_parallel as tp
from .attn_backend import HipAttnBackend
from .cache_manager import PagedKVCacheManager


@dataclass
class LlamaArgs:
    """Fields of chitu/config/models/Meta-Llama-3-8B-Instruct-original.yaml:6-14 (defaults = Llama-3-8B)."""

    dim: int = 4096
    n_layers: int = 32
    n_heads: int = 32
    n_kv_heads: int = 8
    vocab_size: int = 128256
    ffn_dim: int = 14336
    norm_eps: float = 1e-5
    rope_theta: float = 500000.0

    @property
    def head_dim(self):
        return self.dim // self.n_heads


def precompute_freqs_cis(head_dim: int, max_pos: int, theta: float):
    """cos/sin [max_pos, head_dim/2] fp32 (models/model.py:81-88: precompute_freqs_cis; torch.polar like the
    reference so the table is bit-identical)."""
    freqs = 1.0 / (theta ** (torch.arange(0, head_dim, 2)[: head_dim // 2].float() / head_dim))
    ang = torch.outer(torch.arange(max_pos, dtype=torch.float32), freqs)
    cis = torch.polar(torch.ones_like(ang), ang)
    return cis.real.contiguous(), cis.imag.contiguous()


def _param(*shape, device=None):
    return torch.nn.Parameter(torch.empty(*shape, dtype=torch.bfloat16, device=device), requires_grad=False)


# Decode batches up to this size run a layer's two residual-add + RMSNorm steps as the prologue of the GEMMs behind
# them, and RoPE + the K / V page append as the epilogue of the qkv projection (ops.bf16_linear_add_norm[_qkv_post] /
# bf16_linear_silu_add_norm: bit-identical, three launches fewer per layer).  Measured on Llama-3-8B (tools/llama_ab.py,
# profiles/r02_ab_llama_norm_prologue.txt): bs 1 3.45 -> 3.20 ms/step, bs 2 3.57 -> 3.44, bs 4 3.70 -> 3.84 (the
# four-row prologue costs more registers and LDS traffic than the launches it saves).  0 = off.
FUSE_NORM_MAX_BS = int(os.environ.get("CHITU_FUSE_NORM_MAX_BS", "2"))


def _fuses_norm(x, pending, n_out, two_terms_ok: bool = False) -> bool:
    """pending must be a plain [bs, dim] tensor: not None (first layer), not a partial whose all-reduce the norm launch
    itself performs (tensor_parallel.PendingAllReduce, the in-graph xGMI form).  two_terms_ok: also [bs, 2, dim] (a top-2 MoE's
    un-summed outputs), for the prologue that takes them (ops.bf16_linear_add_norm)."""
    dims_ok = isinstance(pending, torch.Tensor) and (pending.dim() == 2 or (two_terms_ok and pending.dim() == 3 and pending.shape[1] == 2))
    return dims_ok and x.shape[0] <= FUSE_NORM_MAX_BS and ops.bf16_add_norm_fits(x.shape[0], n_out, x.shape[1])


class LlamaAttention(torch.nn.Module):
    def __init__(self, args, layer_id, cache, attn_backend, device=None, rotary_type="llama"):
        super().__init__()
        self.rotary_type = rotary_type  # "llama" (Meta original, interleaved) | "hf-llama" (half split)
        t = tp.get_tp_size()
        self.layer_id, self.cache, self.attn_backend = layer_id, cache, attn_backend
        self.hd = args.head_dim
        self.hq, self.hkv = args.n_heads // t, args.n_kv_heads // t
        # merged [wq | wk | wv] rows (ColumnParallel: heads split across ranks), wo RowParallel
        self.wqkv = _param((self.hq + 2 * self.hkv) * self.hd, args.dim, device=device)
        self.wo = _param(args.dim, self.hq * self.hd, device=device)

    def decode_forward_paged(self, x, cos, sin):
        """x = attn_norm(h) bf16 [bs, dim] -> wo(attention) before the all-reduce (model.py:167-198)."""
        return self.decode_from_qkv(ops.bf16_linear(x, self.wqkv), cos, sin)

    def decode_from_qkv(self, qkv, cos, sin):
        """The merged projection's output [bs, (hq + 2 hkv) * hd] -> wo(attention) before the all-reduce."""
        bs = qkv.shape[0]
        qkv = qkv.view(bs, self.hq + 2 * self.hkv, self.hd)
        k_cache, v_cache = self.cache.get_paged_kv_cache(self.layer_id)
        # RoPE(q, k) + append of k and v to their pages: one launch
        q = ops.gqa_qkv_post(qkv, self.hq, self.hkv, cos, sin, k_cache, v_cache, self.cache.get_gpu_block_table(),
                             self.cache.get_gpu_seq_lens_excl_this_decode(), rotary_type=self.rotary_type)
        return self.decode_from_q(q)

    def decode_from_q(self, q):
        """Rotated q heads [bs, hq, hd] (this token's k and v already in their pages) -> wo(attention)."""
        bs = q.shape[0]
        k_cache, v_cache = self.cache.get_paged_kv_cache(self.layer_id)
        table = self.cache.get_gpu_block_table()
        o = self.attn_backend.attn_with_kvcache(
            q.unsqueeze(1), k_cache, v_cache, None, None,
            cache_seqlens=self.cache.get_gpu_seq_lens_incl_this_decode()[:bs], block_table=table[:bs])
        return ops.bf16_linear(o.view(bs, self.hq * self.hd), self.wo)

    def decode_from_residual(self, x, pending, norm_weight, eps, cos, sin):
        """Small decode batches: residual add + attn_norm + wqkv projection [+ RoPE and the K / V page append when the
        rotary pairs are interleaved] in ONE launch.  Returns (x + pending, wo(attention) before the all-reduce)."""
        if self.rotary_type == "llama":
            k_cache, v_cache = self.cache.get_paged_kv_cache(self.layer_id)
            x, qkv = ops.bf16_linear_add_norm_qkv_post(
                x, pending, norm_weight, eps, self.wqkv, self.hq, self.hkv, cos, sin, k_cache, v_cache,
                self.cache.get_gpu_block_table(), self.cache.get_gpu_seq_lens_excl_this_decode())
            return x, self.decode_from_q(qkv[:, : self.hq])
        x, qkv = ops.bf16_linear_add_norm(x, pending, norm_weight, eps, self.wqkv)
        return x, self.decode_from_qkv(qkv, cos, sin)

    def prefill_forward(self, x, cos, sin, varlens):
        """models/model.py:104-132: projections on all T prompt tokens, RoPE, page writes by the cache
        manager (cache_manager.py:93-142), causal GQA through attn_backend.attn_varlen_func."""
        T = x.shape[0]
        qkv = ops.bf16_linear(x, self.wqkv).view(T, self.hq + 2 * self.hkv, self.hd)
        q, k = ops.apply_rotary_pos_emb(qkv[:, : self.hq], qkv[:, self.hq : self.hq + self.hkv], cos, sin, rotary_type=self.rotary_type)
        v = qkv[:, self.hq + self.hkv :].contiguous()
        self.cache.finalize_cache_bylayer_prefill(k, v, self.cache.curr_req_ids, self.cache.curr_varlens, self.layer_id)
        o = self.attn_backend.attn_varlen_func(q, k, v, varlens.prefix_lens, varlens.prefix_lens, varlens.max_len,
                                               varlens.max_len, causal=True)
        return ops.bf16_linear(o.reshape(T, self.hq * self.hd), self.wo)


class LlamaFeedForward(torch.nn.Module):
    """w2(silu(w1 x) * w3 x) with w1 / w3 merged row-wise (model.py:201-214)."""

    def __init__(self, args: LlamaArgs, device=None):
        super().__init__()
        t = tp.get_tp_size()
        self.inter = args.ffn_dim // t
        self.w13 = _param(2 * self.inter, args.dim, device=device)
        self.w2 = _param(args.dim, self.inter, device=device)

    def forward(self, x):
        return ops.bf16_linear(ops.bf16_linear_silu(x, self.w13), self.w2)


class LlamaBlock(torch.nn.Module):
    def __init__(self, layer_id, args, cache, attn_backend, device=None):
        super().__init__()
        self.attn = LlamaAttention(args, layer_id, cache, attn_backend, device)
        self.ffn = LlamaFeedForward(args, device)
        self.attn_norm = _param(args.dim, device=device)
        self.ffn_norm = _param(args.dim, device=device)
        self.eps = args.norm_eps

    def forward(self, x, pending, cos, sin, varlens=None):
        """(x, pending) -> (x', pending'): residual adds folded into the RMSNorm that consumes them;
        varlens given = prefill."""
        if varlens is None and _fuses_norm(x, pending, self.attn.wqkv.shape[0]):
            # small decode batches: the add + norm run as the prologue of the projection that consumes them
            x, a = self.attn.decode_from_residual(x, pending, self.attn_norm, self.eps, cos, sin)
            a = tp.defer_all_reduce(a)
        else:
            x, hn = tp.add_norm(x, pending, self.attn_norm, self.eps)[:2]
            if varlens is None:
                a = tp.defer_all_reduce(self.attn.decode_forward_paged(hn, cos, sin))
            else:
                a = tp.defer_all_reduce(self.attn.prefill_forward(hn, cos, sin, varlens))
        return self.ffn_part(x, a, varlens)

    def ffn_part(self, x, a, varlens):
        if varlens is None and _fuses_norm(x, a, self.ffn.inter):
            x, h = ops.bf16_linear_silu_add_norm(x, a, self.ffn_norm, self.eps, self.ffn.w13)
            return x, tp.defer_all_reduce(ops.bf16_linear(h, self.ffn.w2))
        x, hn = tp.add_norm(x, a, self.ffn_norm, self.eps)[:2]
        return x, tp.defer_all_reduce(self.ffn(hn))


class LlamaDecoder(torch.nn.Module):
    """embed -> blocks -> norm -> head -> fp32 logits; one hipGraph per batch size (model.py:538-622)."""

    block_type = None  # subclasses (Mixtral) swap the block

    def __init__(self, args: LlamaArgs, cache: PagedKVCacheManager, attn_backend: HipAttnBackend,
                 max_position_embeddings: int = 4096, device="cuda"):
        super().__init__()
        self.args, self.cache, self.attn_backend, self.device = args, cache, attn_backend, torch.device(device)
        t = tp.get_tp_size()
        assert args.vocab_size % t == 0 and args.n_kv_heads % t == 0 and args.head_dim == 128, "gqa_decode: head_dim 128"
        self.vocab_local = args.vocab_size // t
        self.vocab_start = tp.get_tp_rank() * self.vocab_local
        self.embed_weight = _param(self.vocab_local, args.dim, device=device)
        block = self.block_type or LlamaBlock
        self.layers = torch.nn.ModuleList(block(i, args, cache, attn_backend, device) for i in range(args.n_layers))
        self.norm = _param(args.dim, device=device)
        self.head_weight = _param(self.vocab_local, args.dim, device=device)
        cos, sin = precompute_freqs_cis(args.head_dim, max_position_embeddings, args.rope_theta)
        self.cos_table, self.sin_table = cos.to(device), sin.to(device)
        self.graphs, self.static_tokens, self.static_out = {}, {}, {}
        self.graph_pool = None

    def embed(self, tokens):
        if self.vocab_local == self.args.vocab_size:
            return F.embedding(tokens, self.embed_weight)
        local = tokens - self.vocab_start
        mask = (local < 0) | (local >= self.vocab_local)
        y = F.embedding(torch.where(mask, torch.zeros_like(local), local), self.embed_weight)
        return tp.all_reduce(torch.where(mask.unsqueeze(-1), torch.zeros_like(y), y))

    @torch.inference_mode()
    def prefill(self, tokens, req_ids):
        """Ragged prompts -> fp32 logits [n_req, vocab] of each prompt's last token; fills the KV pages
        (prefill_single_device, model.py:451-465)."""
        from .deepseek_v3 import VarLens

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
        h = ops.rms_norm(h[last], self.norm, self.args.norm_eps, add=tp.resolve(pending)[last].contiguous())[1]
        self.cache.finalize_cache_all_prefill(req_ids, varlens)
        return tp.all_gather_last_dim(ops.bf16_linear(h, self.head_weight)).float()

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

    def decode_eager(self, tokens):
        # embedding rows of this rank's vocabulary slice + every sequence's rotary row: one launch
        h, cos, sin = ops.embed_rope_gather(tokens, self.embed_weight, self.vocab_start if self.vocab_local != self.args.vocab_size else 0,
                                            self.cache.get_gpu_seq_lens_excl_this_decode(), self.cos_table, self.sin_table)
        if self.vocab_local != self.args.vocab_size:
            h = tp.all_reduce(h)
        pending = None
        for layer in self.layers:
            h, pending = layer(h, pending, cos, sin)
        h = tp.add_norm(h, pending, self.norm, self.args.norm_eps)[1]
        return tp.all_gather_last_dim(ops.bf16_linear(h, self.head_weight), out_dtype=torch.float32)

    @torch.inference_mode()
    def decode(self, tokens, use_graph=True):
        """Eager, or the step captured per batch size: one hipGraph, or -- on a rank with collectives --
        hipGraph pieces with the collectives between them (chitu_amd/graphs.py)."""
        tp.check_comm()  # a collective of an earlier step that timed out: raise instead of decoding garbage
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
            # eager step on the static inputs (also re-writes this step's KV rows), capture, ONE checked replay
            g, self.graph_pool, self.static_out[bs] = graphs.capture_verified(
                lambda: self.decode_eager(self.static_tokens[bs]), self.static_out.get(bs), mode, self.graph_pool,
                what=f"{type(self).__name__} decode step bs={bs}")
            self.graphs[key] = g
        self.graphs[key].replay()
        return self.static_out[bs]


@torch.no_grad()
def init_synthetic_(model: torch.nn.Module, seed: int = 0):
    """bf16 weights randn / sqrt(fan_in) (unit-gain linears), norm weights 1, embeddings randn."""
    dev = next(model.parameters()).device
    gen = torch.Generator(device=dev).manual_seed(seed)
    for name, p in model.named_parameters():
        if name.endswith("norm"):
            p.data.fill_(1.0)
            continue
        std = 1.0 if name == "embed_weight" else p.shape[-1] ** -0.5
        flat = p.data.view(-1)
        for i in range(0, flat.numel(), 1 << 26):
            n = min(1 << 26, flat.numel() - i)
            flat[i : i + n].copy_((torch.randn(n, device=dev, generator=gen) * std).to(p.dtype))
    return model
