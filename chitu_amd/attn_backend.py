"""Attention backend for gfx950 behind the reference's `AttnBackend` interface.

Reference (read-only): chitu/attn_backend.py -- interface AttnBackend:24-164,
TritonAttnBackend:687-774 (the MLA decode path the reference runs on non-NVIDIA devices),
FlashMLABackend:504-572 / FlashInferBackend:575-684 (graph-capturable third-party paths).
`HipAttnBackend` implements `prepare_metadata_for_decode`, `mla_attn_with_kvcache`,
`attn_with_kvcache(block_table=...)` and the MLA-prefill form of `attn_varlen_func` with hand-written HIP kernels; the host side only sizes
the KV split count from the batch (graph-static) and keeps a persistent scratch buffer.
"""

import os
from typing import Optional, Union

import torch

from . import _lib, workspace
from ._lib import check, f32, i32, i64, ptr, require_cuda, stream_ptr
from .ops import append_to_paged_kv_cache

__all__ = ["AttnBackend", "HipAttnBackend"]


# Upper bound of the KV splits of the GQA decode launch (graph-static: sized from the page table's width, not from the
# lengths).  32 by the sweep of round 4 (Llama-3-8B bs 1, ctx 1024: 64 -> 3.064, 32 -> 3.041, 16 -> 3.077, 8 -> 3.175 ms/step;
# from bs 4 on the CU-count term decides; profiles/r04_gqa_split_sweep.txt); CHITU_GQA_MAX_SPLITS overrides it for sweeps.
_GQA_MAX_SPLITS = int(os.environ.get("CHITU_GQA_MAX_SPLITS", "32"))
# single-wave workgroups per CU the split count aims for once batch * kv_heads decides (sweeps: CHITU_GQA_WAVES_PER_CU)
_GQA_WAVES_PER_CU = int(os.environ.get("CHITU_GQA_WAVES_PER_CU", "4"))


class AttnBackend:
    """Interface (chitu/attn_backend.py:24-164)."""

    def prepare_metadata_for_decode(self, *args, **kwargs):
        pass

    def attn_varlen_func(self, q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0,
                         causal=False, window_size=(-1, -1), softcap=0.0, softmax_scale=None):
        raise NotImplementedError()

    def attn_with_kvcache(self, q, k_cache, v_cache, k=None, v=None, cache_seqlens=None, cache_leftpad=None,
                          block_table=None, causal=False, window_size=(-1, -1), softcap=0.0, softmax_scale=None):
        raise NotImplementedError()

    def mla_attn_with_kvcache(self, *args, **kwargs):
        raise NotImplementedError()


def _num_cus():
    try:
        return torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    except Exception:
        return 256


def choose_num_splits(batch: int, head_blocks: int, max_tiles: int, target_wgs: Optional[int] = None) -> int:
    """KV splits so that batch*head_blocks*splits workgroups fill the chip once (the decode kernel
    runs one 96 KB-LDS workgroup per CU), never more splits than 64-token tiles.  Depends only on
    graph-static quantities (batch, max context)."""
    import os

    if os.environ.get("CHITU_MLA_SPLITS"):
        return max(1, min(max_tiles, int(os.environ["CHITU_MLA_SPLITS"])))  # tuning knob
    if target_wgs is None:
        target_wgs = _num_cus()
    per = max(1, batch * head_blocks)
    s = max(1, min(max_tiles, (target_wgs + per - 1) // per))
    return min(s, 64)


_tickets = {}


def _fuse_tickets(device) -> torch.Tensor:
    """The arrival / departure words of chitu_hip_mla_decode_merge_uv_quant_fp8 (+ its sticky error word, the last one): zero
    at creation, left zero by every launch.  One buffer per (device, workspace namespace) for the life of the process --
    launches that share it must not overlap, and captured launches keep a valid address; created on the first EAGER call
    (decode() warms up eagerly before it captures)."""
    key = (torch.device(device).index or 0, workspace._namespace)
    t = _tickets.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("mla_decode_merge_uv_quant must run once eagerly before graph capture")
        import ctypes

        n = ctypes.c_int64(0)
        check(_lib.lib().chitu_hip_mla_decode_tickets_bytes(ctypes.byref(n)), "mla_decode_tickets_bytes")
        t = _tickets[key] = torch.zeros(n.value // 4, dtype=torch.int32, device=device)
    return t


def fused_tail_timed_out(device) -> bool:
    """Did any fused MLA decode launch on this device give up waiting for a split (its 200 ms bound)?  Synchronises."""
    return any(int(t[-1].item()) != 0 for (dev, _), t in _tickets.items() if dev == (torch.device(device).index or 0))


class HipAttnBackend(AttnBackend):
    """MLA absorb-mode paged decode (+ GQA paged decode) on hand-written HIP kernels.

    Construction mirrors TritonAttnBackend (attn_backend.py:688-696) but takes the model
    dimensions explicitly instead of reading hydra globals.
    """

    def __init__(self, local_n_heads: int = 16, kv_lora_rank: int = 512, qk_rope_head_dim: int = 64,
                 qk_nope_head_dim: int = 128, max_seq_len: int = 4096):
        self.local_n_heads = local_n_heads
        self.kv_lora_rank = kv_lora_rank
        self.qk_rope_head_dim = qk_rope_head_dim
        self.qk_nope_head_dim = qk_nope_head_dim
        self.max_seq_len = max_seq_len
        self.block_size = None
        self.num_splits = None

    def prepare_metadata_for_decode(
        self,
        cache_seqlens_excl_this_decode,
        cache_seqlens_incl_this_decode,
        block_table,
        block_size,
        softmax_scale=None,
    ):
        """Runs OUTSIDE the captured graph (model.py:540 -> model_deepseek_v3.py:1339).  Unlike
        FlashInfer's prepare (attn_backend.py:620-637: Python loops with .item() syncs) this does
        no device work: the split count depends only on the batch size and the table width."""
        self.block_size = block_size
        bs = int(cache_seqlens_incl_this_decode.shape[0])
        max_tiles = max(1, (int(block_table.shape[1]) * int(block_size) + 63) // 64)
        self.num_splits = choose_num_splits(bs, (self.local_n_heads + 15) // 16, max_tiles)

    def mla_decode(self, q_nope, q_pe, kv_cache, cache_seqlens_incl, block_table, softmax_scale,
                   num_splits: Optional[int] = None, out: Optional[torch.Tensor] = None,
                   return_partials: bool = False):
        """softmax(scale * (q_nope.c + q_pe.k_pe)) . c over the paged latent cache; no append.

        return_partials=True (fused consumer, ops.mla_merge_absorb_uv_quant_fp8): when the KV range is
        split, skip the merge pass and return (workspace, num_splits) instead of the output; with a
        single split the output tensor is returned as usual."""
        require_cuda(q_nope, q_pe, kv_cache, cache_seqlens_incl, block_table)
        assert kv_cache.ndim == 3 and kv_cache.is_contiguous()  # (num_blocks, block_size, dim)
        assert kv_cache.dtype == torch.bfloat16 and q_nope.dtype == torch.bfloat16 and q_pe.dtype == torch.bfloat16
        assert block_table.dtype == torch.int32 and cache_seqlens_incl.dtype == torch.int32
        assert block_table.stride(1) == 1 and cache_seqlens_incl.is_contiguous()
        B, H, C = q_nope.shape
        R = q_pe.shape[-1]
        assert kv_cache.shape[-1] == C + R

        def ok(t):
            return t.stride(-1) == 1 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0

        if not ok(q_nope):
            q_nope = q_nope.contiguous()
        if not ok(q_pe):
            q_pe = q_pe.contiguous()
        if num_splits is None:
            max_tiles = max(1, (int(block_table.shape[1]) * int(kv_cache.shape[1]) + 63) // 64)
            num_splits = self.num_splits or choose_num_splits(B, (H + 15) // 16, max_tiles)
        partials = return_partials and num_splits > 1
        if out is None and not partials:
            out = torch.empty(B, H, C, dtype=torch.bfloat16, device=q_nope.device)
        need = B * H * num_splits * (C * 2 + 4) if num_splits > 1 else 0  # bf16 partial rows + fp32 LSE
        ws = workspace.get(max(need, 1), q_nope.device, "mla")
        check(
            _lib.lib().chitu_hip_mla_decode(
                ptr(q_nope), i64(q_nope.stride(0)), i64(q_nope.stride(1)), ptr(q_pe), i64(q_pe.stride(0)),
                i64(q_pe.stride(1)), ptr(kv_cache), i64(kv_cache.shape[0]), i32(kv_cache.shape[1]),
                ptr(block_table), i32(block_table.stride(0)), ptr(cache_seqlens_incl), f32(softmax_scale),
                ptr(None if partials else out), i32(B), i32(H), i32(C), i32(R), i32(num_splits), ptr(ws),
                i64(ws.numel()), stream_ptr(),
            ),
            "mla_decode",
        )
        return (ws, num_splits) if partials else out

    def mla_decode_merge_uv_quant(self, q_nope, q_pe, kv_cache, cache_seqlens_incl, block_table, softmax_scale, w_uv, scale,
                                  scale_offset, scale_stride_h, scale_stride_k, num_splits: Optional[int] = None,
                                  tile_major: bool = False):
        """mla_decode(return_partials=True) + ops.mla_merge_absorb_uv_quant_fp8 in ONE launch (round 6,
        chitu_hip_mla_decode_merge_uv_quant_fp8): every split workgroup waits for its sequence's other splits and finishes one
        head -- merge, o . W_UV^T (model_deepseek_v3.py:697), act_quant of wo's input.  Bit-identical to the two launches.
        Returns what the merge op returns, or None when the shape takes the two-launch form (one split, > 4096 groups)."""
        from . import ops

        require_cuda(q_nope, q_pe, kv_cache, cache_seqlens_incl, block_table, w_uv, scale)
        assert kv_cache.ndim == 3 and kv_cache.is_contiguous() and kv_cache.dtype == torch.bfloat16
        assert q_nope.dtype == torch.bfloat16 and q_pe.dtype == torch.bfloat16
        assert block_table.dtype == torch.int32 and cache_seqlens_incl.dtype == torch.int32
        assert block_table.stride(1) == 1 and cache_seqlens_incl.is_contiguous()
        B, H, C = q_nope.shape
        R = q_pe.shape[-1]
        assert kv_cache.shape[-1] == C + R and C == 512
        assert w_uv.element_size() == 1 and scale.dtype == torch.float32 and w_uv.dim() == 3 and tuple(w_uv.shape) == (H, 128, C)
        assert w_uv.stride(2) == 1 and w_uv.stride(1) == C

        def ok(t):
            return t.stride(-1) == 1 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0

        if not ok(q_nope):
            q_nope = q_nope.contiguous()
        if not ok(q_pe):
            q_pe = q_pe.contiguous()
        if num_splits is None:
            max_tiles = max(1, (int(block_table.shape[1]) * int(kv_cache.shape[1]) + 63) // 64)
            num_splits = self.num_splits or choose_num_splits(B, (H + 15) // 16, max_tiles)
        if num_splits < 2 or B * ((H + 15) // 16) > 4096 or w_uv.data_ptr() % 16 != 0:
            return None
        ws = workspace.get(B * H * num_splits * (C * 2 + 4), q_nope.device, "mla")
        if tile_major:
            q, s = ops._tiled_buffers(B, H * 128, w_uv.device)
        else:
            q = torch.empty(B, H * 128, dtype=torch.float8_e4m3fn, device=w_uv.device)
            s = torch.empty(B, H, dtype=torch.float32, device=w_uv.device)
        check(
            _lib.lib().chitu_hip_mla_decode_merge_uv_quant_fp8(
                ptr(q_nope), i64(q_nope.stride(0)), i64(q_nope.stride(1)), ptr(q_pe), i64(q_pe.stride(0)), i64(q_pe.stride(1)),
                ptr(kv_cache), i64(kv_cache.shape[0]), i32(kv_cache.shape[1]), ptr(block_table), i32(block_table.stride(0)),
                ptr(cache_seqlens_incl), f32(softmax_scale), i32(B), i32(H), i32(C), i32(R), i32(num_splits), ptr(ws),
                i64(ws.numel()), ptr(w_uv), i64(w_uv.stride(0)), ptr(scale), i64(scale_offset), i64(scale_stride_h),
                i64(scale_stride_k), ptr(q), ptr(s), i32(1 if tile_major else 0), ptr(_fuse_tickets(q_nope.device)), stream_ptr(),
            ),
            "mla_decode_merge_uv_quant_fp8",
        )
        return (ops.TiledQuant(q, s, B, H * 128), None) if tile_major else (q, s)

    def mla_attn_with_kvcache(
        self,
        q_nope,
        q_pe,
        kv_cache,
        kv,
        cache_seqlens_excl_this_decode: Union[(int, torch.Tensor)],
        cache_seqlens_incl_this_decode: Union[(int, torch.Tensor)],
        block_table: torch.Tensor,
        causal=True,
        window_size=(-1, -1),  # -1 means infinite context window
        softcap=0.0,  # 0.0 means deactivated
        softmax_scale=None,
    ):
        """Same contract as TritonAttnBackend.mla_attn_with_kvcache (attn_backend.py:707-774):
        append this token's [kv_c | k_pe] row to its page, then attend over the sequence.
        Returns [B, 1, H, kv_lora_rank]."""
        assert window_size == (-1, -1) and softcap == 0.0, "not used by the MLA decode path"
        if softmax_scale is None:
            # the reference's default here is accidentally a 1-tuple (attn_backend.py:756-758)
            softmax_scale = 1.0 / ((self.qk_rope_head_dim + self.qk_nope_head_dim) ** 0.5)
        append_to_paged_kv_cache(kv_cache, block_table, kv, cache_seqlens_excl_this_decode)
        B = q_nope.shape[0]
        o = self.mla_decode(q_nope, q_pe, kv_cache, cache_seqlens_incl_this_decode, block_table, float(softmax_scale))
        return o.view(B, 1, q_nope.shape[1], -1)

    # ------------------------------------------------------------------ MLA prefill (absorb mode, MQA 576/512)
    def attn_varlen_func(
        self,
        q,
        k,
        v,
        cu_seqlens_q,
        cu_seqlens_k,
        max_seqlen_q,
        max_seqlen_k,
        dropout_p=0.0,
        causal=False,
        window_size=(-1, -1),
        softcap=0.0,
        softmax_scale=None,
    ):
        """Causal varlen self-attention in the shape AttentionDeepSeekV3.prefill_forward calls it in
        absorb mode (model_deepseek_v3.py:589-599; interface attn_backend.py:39-90): MQA with
        q [T, H, C+R], k [T, 1, C+R] = [kv_norm(kv_c) | rope(k_pe)], v [T, 1, C] = the latent part of k
        (the MLA identity -- v is NOT read, k[..., :C] is used), cu_seqlens_q == cu_seqlens_k.

        chitu_hip_mla_prefill_flash (default): 128 Q rows (8 tokens x 16 heads) per workgroup against every 64-key tile,
        32x32x16 MFMA, Q in registers -- within the attention bar of the decode kernel, not bit for bit.
        CHITU_MLA_PREFILL=exact: chitu_hip_mla_prefill, the decode kernel's tile machinery with four query tokens per
        workgroup; per query token the arithmetic is the decode step's, so prefill and token-by-token decode agree bit
        for bit (3x slower at 2048 tokens; the cross-check).  CHITU_MLA_PREFILL=compose: the kernel-free composition
        (keys staged into 64-token pages, every query token run as one decode "sequence" over them) -- equal to "exact"
        bit for bit, KV re-read once per token.
        No host sync; not graph-captured (prefill never is, model.py:538-546)."""
        assert causal and dropout_p == 0.0 and tuple(window_size) == (-1, -1) and softcap == 0.0
        require_cuda(q, k, cu_seqlens_q, cu_seqlens_k)
        T, H, Dq = q.shape
        C, R = self.kv_lora_rank, self.qk_rope_head_dim
        if Dq == 128 and k.dim() == 3 and k.shape[-1] == 128 and tuple(v.shape) == tuple(k.shape):
            return self._gqa_varlen_causal(q, k, v, cu_seqlens_k, int(max_seqlen_k), softmax_scale)
        assert Dq == C + R and tuple(k.shape) == (T, 1, Dq) and v.shape[0] == T and v.shape[-1] == C, \
            "only the MLA absorb-mode MQA shape (q/k 576, v 512) is implemented"
        assert cu_seqlens_q.shape == cu_seqlens_k.shape and max_seqlen_q == max_seqlen_k
        assert q.dtype == torch.bfloat16 and k.dtype == torch.bfloat16
        if softmax_scale is None:
            softmax_scale = Dq ** -0.5
        if T == 0:
            return q.new_empty(0, H, C)
        dev = q.device
        mode = os.environ.get("CHITU_MLA_PREFILL", "flash")
        assert mode in ("flash", "exact", "compose"), mode
        if mode != "compose":
            kq = q if (q.stride(-1) == 1 and q.stride(0) % 8 == 0 and q.stride(1) % 8 == 0 and q.data_ptr() % 16 == 0) else q.contiguous()
            kk = k.reshape(T, Dq)
            if not (kk.stride(-1) == 1 and kk.stride(0) % 8 == 0 and kk.data_ptr() % 16 == 0):
                kk = kk.contiguous()
            cu32 = cu_seqlens_k.to(device=dev, dtype=torch.int32).contiguous()
            out = torch.empty(T, H, C, dtype=torch.bfloat16, device=dev)
            entry = _lib.lib().chitu_hip_mla_prefill_flash if mode == "flash" else _lib.lib().chitu_hip_mla_prefill
            check(
                entry(
                    ptr(kq), i64(kq.stride(0)), i64(kq.stride(1)), ptr(kk), i64(kk.stride(0)), ptr(cu32),
                    i32(cu32.numel() - 1), i32(int(max_seqlen_k)), f32(softmax_scale), ptr(out), i32(H), i32(C), i32(R),
                    stream_ptr(),
                ),
                "mla_prefill",
            )
            return out
        page = 64
        n_seq = cu_seqlens_k.numel() - 1
        cu = cu_seqlens_k.to(device=dev, dtype=torch.long)
        pos = torch.arange(T, device=dev)
        seq = torch.searchsorted(cu[1:].contiguous(), pos, right=True).clamp_(max=n_seq - 1)
        off = pos - cu[seq]
        pages_per = (cu[1:] - cu[:-1] + page - 1) // page
        base = torch.cumsum(pages_per, 0) - pages_per
        num_pages = T // page + n_seq + 1  # upper bound known on the host
        max_pages = max(1, (int(max_seqlen_k) + page - 1) // page)
        staged = torch.empty(num_pages, page, Dq, dtype=torch.bfloat16, device=dev)
        staged.view(-1, Dq).index_copy_(0, base[seq] * page + off, k.reshape(T, Dq))
        table = (base[seq].unsqueeze(1) + torch.arange(max_pages, device=dev)).clamp_(max=num_pages - 1).to(torch.int32)
        lens = (off + 1).to(torch.int32)
        out = torch.empty(T, H, C, dtype=torch.bfloat16, device=dev)
        step = 32768  # grid.y limit of one launch
        for t0 in range(0, T, step):
            t1 = min(T, t0 + step)
            self.mla_decode(q[t0:t1, :, :C], q[t0:t1, :, C:], staged, lens[t0:t1].contiguous(), table[t0:t1].contiguous(),
                            softmax_scale, num_splits=1, out=out[t0:t1])
        return out

    def _gqa_varlen_causal(self, q, k, v, cu_seqlens, max_seqlen, softmax_scale):
        """Causal GQA / MHA prefill attention (Attention.prefill_forward, models/model.py:104-132), head_dim 128:
        q [T, Hq, 128], k / v [T, Hkv, 128].  Default: the flash kernel chitu_hip_gqa_prefill.  CHITU_GQA_PREFILL=compose
        (and shapes the kernel does not take): the round-2 composition from the decode kernel -- keys / values staged once
        into 256-token pages, every query token one decode "sequence" of length pos + 1 over its own sequence's pages
        (chitu_hip_gqa_decode): exact causal attention with the decode numerics, KV re-read once per query token; kept
        as the cross-check.  No host sync either way."""
        T, Hq, D = q.shape
        Hkv = k.shape[1]
        if T == 0:
            return q.new_empty(0, Hq, D)
        dev = q.device
        G = Hq // Hkv
        if (os.environ.get("CHITU_GQA_PREFILL", "flash") != "compose" and D == 128 and Hq % Hkv == 0 and G <= 32
                and (G & (G - 1)) == 0):
            # chitu_hip_gqa_prefill (csrc/gqa_prefill_flash.hip): 128 Q rows (128 / G tokens x G heads of one KV head) per
            # workgroup against 64-key K / V tiles; KV read once per 128 / G query tokens instead of once per token
            def ok(t):
                return t.stride(-1) == 1 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0 and t.data_ptr() % 16 == 0

            qq, kk, vv = (t if ok(t) else t.contiguous() for t in (q, k, v))
            cu32 = cu_seqlens.to(device=dev, dtype=torch.int32).contiguous()
            out = torch.empty(T, Hq, D, dtype=torch.bfloat16, device=dev)
            if softmax_scale is None:
                softmax_scale = D ** -0.5
            check(_lib.lib().chitu_hip_gqa_prefill(
                ptr(qq), i64(qq.stride(0)), i64(qq.stride(1)), ptr(kk), i64(kk.stride(0)), i64(kk.stride(1)), ptr(vv),
                i64(vv.stride(0)), i64(vv.stride(1)), ptr(cu32), i32(cu32.numel() - 1), i32(int(max_seqlen)), f32(softmax_scale),
                ptr(out), i32(Hq), i32(Hkv), i32(D), stream_ptr()), "gqa_prefill")
            return out
        page = 256
        n_seq = cu_seqlens.numel() - 1
        cu = cu_seqlens.to(device=dev, dtype=torch.long)
        pos = torch.arange(T, device=dev)
        seq = torch.searchsorted(cu[1:].contiguous(), pos, right=True).clamp_(max=n_seq - 1)
        off = pos - cu[seq]
        pages_per = (cu[1:] - cu[:-1] + page - 1) // page
        base = torch.cumsum(pages_per, 0) - pages_per
        num_pages = T // page + n_seq + 1
        max_pages = max(1, (max_seqlen + page - 1) // page)
        dst = base[seq] * page + off
        k_pages = torch.zeros(num_pages, page, Hkv, D, dtype=torch.bfloat16, device=dev)
        v_pages = torch.zeros(num_pages, page, Hkv, D, dtype=torch.bfloat16, device=dev)
        k_pages.view(-1, Hkv, D).index_copy_(0, dst, k.reshape(T, Hkv, D))
        v_pages.view(-1, Hkv, D).index_copy_(0, dst, v.reshape(T, Hkv, D))
        table = (base[seq].unsqueeze(1) + torch.arange(max_pages, device=dev)).clamp_(max=num_pages - 1).to(torch.int32)
        lens = (off + 1).to(torch.int32)
        out = torch.empty(T, Hq, D, dtype=torch.bfloat16, device=dev)
        step = 16384
        for t0 in range(0, T, step):
            t1 = min(T, t0 + step)
            out[t0:t1] = self.attn_with_kvcache(q[t0:t1].unsqueeze(1), k_pages, v_pages, None, None, cache_seqlens=lens[t0:t1].contiguous(),
                                                block_table=table[t0:t1].contiguous(), softmax_scale=softmax_scale).view(t1 - t0, Hq, D)
        return out

    # ------------------------------------------------------------------ non-MLA (GQA / MHA) paged decode
    def attn_with_kvcache(
        self,
        q,
        k_cache,
        v_cache,
        k=None,
        v=None,
        cache_seqlens: Optional[Union[(int, torch.Tensor)]] = None,
        cache_leftpad: Optional[torch.Tensor] = None,
        block_table: Optional[torch.Tensor] = None,
        causal=False,
        window_size=(-1, -1),
        softcap=0.0,
        softmax_scale=None,
        num_splits: Optional[int] = None,
    ):
        """Paged single-token attention with in-place append, the contract of
        AttnBackend.attn_with_kvcache (chitu/attn_backend.py:92-164) as FlashAttnBackend implements
        it (:208-243): if k/v are given they are written at position cache_seqlens[b] of each
        sequence's pages, then q attends to cache_seqlens[b] + 1 tokens.

        q [bs, 1, Hq, 128]; k_cache / v_cache [pages, page_size, Hkv, 128]; k / v [bs, 1, Hkv, 128];
        cache_seqlens [bs] int32 (excluding this token); block_table [bs, max_pages] int32.
        Returns [bs, 1, Hq, 128].  (RefAttnBackend rejects block_table, :473 -- this is the paged path.)
        """
        assert block_table is not None, "HipAttnBackend.attn_with_kvcache is the paged path"
        assert cache_leftpad is None and window_size == (-1, -1) and softcap == 0.0
        assert q.dim() == 4 and q.shape[1] == 1, "decode: one query token per sequence"
        require_cuda(q, k_cache, v_cache, cache_seqlens, block_table)
        assert q.dtype == torch.bfloat16 and k_cache.dtype == torch.bfloat16 and v_cache.dtype == torch.bfloat16
        assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.shape == v_cache.shape
        assert cache_seqlens.dtype == torch.int32 and block_table.dtype == torch.int32 and block_table.stride(1) == 1
        bs, _, Hq, D = q.shape
        Hkv = k_cache.shape[2]
        if softmax_scale is None:
            softmax_scale = D ** -0.5
        seqlens = cache_seqlens
        if k is not None:
            assert v is not None
            append_to_paged_kv_cache(k_cache, block_table, k.contiguous(), cache_seqlens)
            append_to_paged_kv_cache(v_cache, block_table, v.contiguous(), cache_seqlens)
            seqlens = cache_seqlens + 1
        q3 = q.view(bs, Hq, D)
        if not (q3.stride(-1) == 1 and q3.stride(0) % 8 == 0 and q3.stride(1) % 8 == 0 and q3.data_ptr() % 16 == 0):
            q3 = q3.contiguous()
        if num_splits is None:
            max_steps = max(1, int(block_table.shape[1]) * int(k_cache.shape[1]) // 16)
            num_splits = max(1, min(max_steps, _GQA_MAX_SPLITS, (_GQA_WAVES_PER_CU * _num_cus()) // max(1, bs * Hkv)))
        out = torch.empty(bs, Hq, D, dtype=torch.bfloat16, device=q.device)
        need = bs * Hq * num_splits * (D + 1) * 4 if num_splits > 1 else 1
        ws = workspace.get(need, q.device, "gqa")
        check(
            _lib.lib().chitu_hip_gqa_decode(
                ptr(q3), i64(q3.stride(0)), i64(q3.stride(1)), ptr(k_cache), ptr(v_cache), i64(k_cache.shape[0]),
                i32(k_cache.shape[1]), i32(Hkv), ptr(block_table), i32(block_table.stride(0)), ptr(seqlens),
                f32(softmax_scale), ptr(out), i32(bs), i32(Hq), i32(D), i32(num_splits), ptr(ws), i64(ws.numel()),
                stream_ptr(),
            ),
            "gqa_decode",
        )
        return out.view(bs, 1, Hq, D)
