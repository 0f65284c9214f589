"""Paged KV cache manager: host page allocator + persistent device buffers.

Mirrors chitu/cache_manager.py:12-225 (PagedKVCacheManager): same methods, same buffers
(`curr_seq_lens_gpu_{excl,incl}_this_decode`, `gpu_block_table_buffer`, `paged_kv_cache` /
`paged_k_cache`+`paged_v_cache`).  Differences, all host-side: the free list is a deque instead
of `list(set)[0]` (the reference's own TODO, :69), per-step H2D traffic is ONE small asynchronous copy from
pinned memory (the two length vectors; the block table travels only on the steps that change it) instead of
one blocking copy per request, and no Timers (their cuda syncs, global_vars.py:132,140, would serialise the
decode loop) -- so the host can prepare step t + 1 while the GPU runs step t.  Device memory layout is unchanged,
so the HIP kernels see exactly the reference's cache.
"""

from collections import deque
from logging import getLogger

import torch

logger = getLogger(__name__)
_BLOCK_SIZE = 512
_MAX_SEQ_LEN = 2048
_STAGE_RING = 4


class PagedKVCacheManager:
    def __init__(
        self,
        begin_layer_id,
        end_layer_id,
        num_hot_req=16,
        block_size=_BLOCK_SIZE,
        max_seq_len=_MAX_SEQ_LEN,
        device="cuda",
        *,
        k_shape_per_sample=None,
        v_shape_per_sample=None,
        kv_shape_per_sample=None,
        n_local_kv_heads=None,
        head_dim=None,
        dtype=None,
    ):
        self.max_blocks_per_req = max_seq_len // block_size + 1
        self.num_blocks = self.max_blocks_per_req * num_hot_req
        self.begin_layer_id = begin_layer_id
        self.end_layer_id = end_layer_id
        self.num_layers = end_layer_id - begin_layer_id
        self.k_shape_per_sample = (
            k_shape_per_sample if k_shape_per_sample is not None else (n_local_kv_heads, head_dim)
        )
        self.v_shape_per_sample = (
            v_shape_per_sample if v_shape_per_sample is not None else (n_local_kv_heads, head_dim)
        )
        self.kv_shape_per_sample = kv_shape_per_sample
        self.block_size = block_size
        self.max_seq_len = max_seq_len
        self.device = torch.device(device)
        self.gpu_block_table = None
        dtype = dtype or torch.get_default_dtype()

        self.seq_lens = {}
        self.block_table = {}  # req_id -> [block ids]
        self.curr_seq_lens = []
        # excl / incl live in one device allocation so that one copy refreshes both (same names and shapes as the reference's)
        self._lens_dev = torch.zeros(2, num_hot_req, dtype=torch.int32, device=self.device)
        self.curr_seq_lens_gpu_excl_this_decode = self._lens_dev[0]
        self.curr_seq_lens_gpu_incl_this_decode = self._lens_dev[1]
        self.gpu_block_table_buffer = torch.zeros(
            (num_hot_req, self.max_blocks_per_req), dtype=torch.int32, device=self.device
        )
        # Host staging: a small RING of pinned buffers + copy_(non_blocking=True).  A pageable copy_() returns only after
        # the runtime has staged the bytes, which waits for the stream -- the host then never runs ahead of the GPU.
        # From pinned memory the copy is just enqueued; a slot is rewritten _STAGE_RING steps later, after the event
        # recorded behind its copy has completed (it has, long since, unless the host is that far ahead).
        self._pinned = self.device.type == "cuda"
        self._host_lens = torch.zeros(_STAGE_RING, 2, num_hot_req, dtype=torch.int32, pin_memory=self._pinned)
        self._host_table = torch.zeros((_STAGE_RING, num_hot_req, self.max_blocks_per_req), dtype=torch.int32,
                                       pin_memory=self._pinned)
        self._lens_np, self._table_np = self._host_lens.numpy(), self._host_table.numpy()
        self._stage_events = {"lens": [None] * _STAGE_RING, "table": [None] * _STAGE_RING}
        self._stage_next = {"lens": 0, "table": 0}
        self._table_rows = None  # request ids whose rows the device table currently holds, in order (None = stale)
        self.free_blocks = deque(range(self.num_blocks))
        if self.kv_shape_per_sample is not None:
            self.paged_kv_cache = torch.zeros(
                (self.num_layers, self.num_blocks, block_size) + tuple(self.kv_shape_per_sample),
                device=device, dtype=dtype,
            )
        else:
            self.paged_k_cache = torch.zeros(
                (self.num_layers, self.num_blocks, block_size) + tuple(self.k_shape_per_sample),
                device=device, dtype=dtype,
            )
            self.paged_v_cache = torch.zeros(
                (self.num_layers, self.num_blocks, block_size) + tuple(self.v_shape_per_sample),
                device=device, dtype=dtype,
            )

    def get_block_size(self):
        return self.block_size

    # ---- prefill: write whole pages (chitu/cache_manager.py:93-142)
    def finalize_cache_bylayer_prefill(self, xk, xv, req_ids, varlen, layer_id):
        """Write the prompt tokens' rows of one layer into their pages (cache_manager.py:93-142).  The pages are taken and
        the destination row of every token is worked out once, at the first layer; each layer is then ONE scatter
        (index_copy_) per cache instead of a slice copy per page."""
        if layer_id == self.begin_layer_id:
            rows = []
            for idx, req_id in enumerate(req_ids):
                n_prepared = (varlen.cpu_lens[idx] + self.block_size - 1) // self.block_size
                self.seq_lens[req_id] = varlen.cpu_lens[idx]
                self.block_table[req_id] = [self.get_free_block() for _ in range(n_prepared)]
                self._table_rows = None
                n_tok = varlen.cpu_prefix_lens[idx + 1] - varlen.cpu_prefix_lens[idx]
                t = torch.arange(n_tok, dtype=torch.int64)
                blocks = torch.tensor(self.block_table[req_id], dtype=torch.int64)
                rows.append(blocks[t // self.block_size] * self.block_size + t % self.block_size)
            dev = (self.paged_kv_cache if self.kv_shape_per_sample is not None else self.paged_k_cache).device
            self._prefill_rows = (torch.cat(rows) if rows else torch.zeros(0, dtype=torch.int64)).to(dev)
        li = layer_id - self.begin_layer_id
        rows = self._prefill_rows
        if self.kv_shape_per_sample is not None:
            c = self.paged_kv_cache[li]
            c.view(-1, *c.shape[2:]).index_copy_(0, rows, xk[: rows.numel()].to(c.dtype))
        else:
            ck, cv = self.paged_k_cache[li], self.paged_v_cache[li]
            ck.view(-1, *ck.shape[2:]).index_copy_(0, rows, xk[: rows.numel()].to(ck.dtype))
            cv.view(-1, *cv.shape[2:]).index_copy_(0, rows, xv[: rows.numel()].to(cv.dtype))

    def register_sequence(self, req_id, length):
        """Allocate pages for a sequence of `length` cached tokens without writing data
        (synthetic benchmarks / tests; prefill normally does this)."""
        self.seq_lens[req_id] = length
        n = (length + self.block_size - 1) // self.block_size
        self.block_table[req_id] = [self.get_free_block() for _ in range(n)]
        self._table_rows = None

    def finalize_cache_all_prefill(self, req_ids, varlen):
        self.curr_varlens = None
        self.curr_req_ids = None

    def _stage_slot(self, kind):
        """Next pinned slot of a ring; waits for the copy that last read it (a no-op unless the host is a ring ahead)."""
        slot = self._stage_next[kind]
        self._stage_next[kind] = (slot + 1) % _STAGE_RING
        ev = self._stage_events[kind][slot]
        if ev is not None:
            ev.synchronize()
        return slot

    def _stage_done(self, kind, slot):
        if self._pinned:
            ev = torch.cuda.Event()
            ev.record()
            self._stage_events[kind][slot] = ev

    def prepare_cache_decode(self, req_ids):
        n = len(req_ids)
        seq_lens = [self.seq_lens[r] for r in req_ids]
        self.curr_seq_lens = seq_lens
        slot = self._stage_slot("lens")
        h = self._lens_np[slot]
        h[0, :n] = seq_lens
        h[1, :n] = h[0, :n] + 1
        # the live columns of both vectors: TWO contiguous row copies from the pinned slot (a strided [2, n] slice of a
        # [2, max] buffer is not one async memcpy for torch: it goes through a pageable temporary and a device copy kernel).
        # The tail keeps its old values: a ring slot's tail holds whatever step last used that slot.
        self._lens_dev[0, :n].copy_(self._host_lens[slot][0, :n], non_blocking=True)
        self._lens_dev[1, :n].copy_(self._host_lens[slot][1, :n], non_blocking=True)
        self._stage_done("lens", slot)

    def get_free_block(self):
        if not self.free_blocks:
            raise Exception("No more free blocks.")  # same behaviour as cache_manager.py:163-164
        return self.free_blocks.popleft()

    def get_gpu_block_table(self):
        return self.gpu_block_table

    def get_gpu_seq_lens_excl_this_decode(self):
        return self.curr_seq_lens_gpu_excl_this_decode[: len(self.curr_seq_lens)]

    def get_gpu_seq_lens_incl_this_decode(self):
        return self.curr_seq_lens_gpu_incl_this_decode[: len(self.curr_seq_lens)]

    def get_paged_kv_cache(self, layer_id):
        if self.kv_shape_per_sample is not None:
            return self.paged_kv_cache[layer_id - self.begin_layer_id]
        return (
            self.paged_k_cache[layer_id - self.begin_layer_id],
            self.paged_v_cache[layer_id - self.begin_layer_id],
        )

    def free_req_cache_blocks(self, req_id):
        for block in self.block_table[req_id]:
            self.free_blocks.append(block)
        del self.block_table[req_id]
        self._table_rows = None

    def prepare_block_table_for_decode(self, req_ids):
        """Make room for the token this decode step appends, then refresh the device table
        (chitu/cache_manager.py:196-209).  Page allocation stays on the host, outside the graph.  The table is copied
        only on the steps that change it (a request crossed a page boundary, or the batch's rows changed): in a
        steady decode that is one step in `block_size`."""
        n = len(req_ids)
        changed = self._table_rows is None or self._table_rows != list(req_ids)
        for req_id in req_ids:
            if self.seq_lens[req_id] % self.block_size == 0:
                self.block_table[req_id].append(self.get_free_block())
                changed = True
        if changed:
            slot = self._stage_slot("table")
            h = self._table_np[slot]
            h[:n] = 0
            for idx, req_id in enumerate(req_ids):
                ids = self.block_table[req_id]
                h[idx, : len(ids)] = ids
            self.gpu_block_table_buffer[:n].copy_(self._host_table[slot, :n], non_blocking=True)
            self._stage_done("table", slot)
            self._table_rows = list(req_ids)
        self.gpu_block_table = self.gpu_block_table_buffer[:n]

    def finalize_cache_single_decode(self, req_ids):
        for req_id in req_ids:
            self.seq_lens[req_id] = self.seq_lens[req_id] + 1
        self.curr_varlens = None
        self.curr_req_ids = None

    def finalize_cache_all_decode(self, req_id):
        assert req_id in self.seq_lens
        assert req_id in self.block_table
        del self.seq_lens[req_id]
        self.free_req_cache_blocks(req_id)
        self.curr_varlens = None
        self.curr_req_ids = None
