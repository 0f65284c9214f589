"""In-graph tensor-parallel collectives over xGMI (host side of csrc/comm.hip).

Reference seam: every collective of the reference's decode step is a `torch.distributed` call on an NCCL
group (chitu/tensor_parallel.py:94-102, :157-169, :199-208; chitu/models/model_deepseek_v3.py:1010-1011)
captured into the step's CUDA graph (chitu/models/model.py:554-617).  Here the same collectives are HIP
kernels that push over xGMI into peer-mapped uncached buffers, so the step stays ONE hipGraph and the
launches around an all-reduce ([top-k sum ->] all-reduce -> residual add -> RMSNorm -> fp8 quant) are one.

`XgmiComm` owns one rank's buffer and the mappings of its peers' buffers.  Ranks in different processes
exchange 64-byte IPC handles over the process group (`XgmiComm.from_group`, any backend: the handles are
host bytes); ranks that share a process (tests) exchange raw pointers (`XgmiComm.connect_local`).
"""

import ctypes
from typing import List, Optional

import torch

from . import _lib
from ._lib import check, f32, i32, i64, ptr, require_cuda, stream_ptr

_QUANT_MODE = {None: 0, "act": 1, "group": 2}
closed_buffers = []  # device addresses of the (uncached) exchange buffers this process has freed: diagnostics (tests/conftest.py)


def _mark_recycled():
    from . import graphs

    graphs.mark_memory_recycled()


class XgmiComm:
    def __init__(self, rank: int, world: int, max_rows: int = 64, max_dim: int = 8192,
                 gather_bytes: int = 0, timeout_ms: int = 10000):
        self.rank, self.world, self.max_rows, self.max_dim, self.gather_bytes = rank, world, max_rows, max_dim, gather_bytes
        self.device = torch.device("cuda", torch.cuda.current_device())
        h = ctypes.c_void_p()
        check(_lib.lib().chitu_hip_comm_create(i32(rank), i32(world), i32(max_rows), i32(max_dim), i64(gather_bytes),
                                               i32(timeout_ms), ctypes.byref(h)), "comm_create")
        self._h = h
        _mark_recycled()  # an uncached exchange buffer came to life: eager launches sweep the L2s once (graphs.py, the canary)
        self.two_shot_bytes = 256 << 10  # the library's default (chitu_hip_comm_set_two_shot)
        # Split-phase mode (tensor_parallel.enable_xgmi under CHITU_XGMI_SPLIT_PHASE=1): a process group over which every
        # whole collective (phase 0) is run as contribute -> host barrier -> complete, so that no kernel ever waits for
        # a peer -- the form in which 4 or 8 rank PROCESSES can share one time-sliced GPU (the single-GPU stand-in for a
        # node: real handle exchange, peer mapping, slicing, epochs; eager launches only).  None = the product form.
        self.split_phase_group = None

    # ------------------------------------------------------------------ wiring
    def ipc_handle(self) -> bytes:
        buf = ctypes.create_string_buffer(64)
        check(_lib.lib().chitu_hip_comm_ipc_handle(self._h, buf), "comm_ipc_handle")
        return buf.raw

    def open_peer(self, peer: int, handle: bytes):
        assert len(handle) == 64
        check(_lib.lib().chitu_hip_comm_open_peer(self._h, i32(peer), ctypes.c_char_p(handle)), "comm_open_peer")
        _mark_recycled()

    def local_ptr(self) -> int:
        p = ctypes.c_void_p()
        check(_lib.lib().chitu_hip_comm_local_ptr(self._h, ctypes.byref(p)), "comm_local_ptr")
        return p.value

    def set_peer(self, peer: int, pointer: int):
        check(_lib.lib().chitu_hip_comm_set_peer(self._h, i32(peer), ctypes.c_void_p(pointer)), "comm_set_peer")

    @staticmethod
    def connect_local(comms: List["XgmiComm"]):
        """Ranks living in ONE process (tests): wire them with raw pointers."""
        for a in comms:
            for b in comms:
                if a is not b:
                    a.set_peer(b.rank, b.local_ptr())

    @classmethod
    def from_group(cls, group=None, **kw) -> "XgmiComm":
        """One rank per process: create, exchange the IPC handles over `group` (host objects, any backend),
        map every peer.  Collective over the group and COLLECTIVE-SAFE: a rank whose own stage failed still takes
        part in every exchange (sending a failure marker), all ranks agree on the outcome of each stage before the
        next one starts, and either every rank returns a comm or every rank raises RuntimeError -- no rank is ever
        left alone inside a collective the others have skipped."""
        import torch.distributed as dist

        rank, world = dist.get_rank(group), dist.get_world_size(group)

        def agree(ok: bool, what: str, why: str = ""):
            votes = [None] * world
            dist.all_gather_object(votes, (bool(ok), why), group=group)
            bad = [(r, w) for r, (o, w) in enumerate(votes) if not o]
            if bad:
                raise RuntimeError(f"xGMI setup failed at '{what}' on rank(s) " + ", ".join(f"{r}: {w}" for r, w in bad))

        comm, handle, why = None, None, ""
        try:
            comm = cls(rank, world, **kw)
            handle = comm.ipc_handle()
        except Exception as e:  # noqa: BLE001 -- reported to every rank below
            why = repr(e)
        handles = [None] * world
        dist.all_gather_object(handles, handle, group=group)  # a failed rank contributes None
        try:
            agree(handle is not None, "create", why)
            why = ""
            try:
                for peer, h in enumerate(handles):
                    if peer != rank:
                        comm.open_peer(peer, h)
            except Exception as e:  # noqa: BLE001
                why = repr(e)
            agree(not why, "open_peer", why)
        except RuntimeError:
            if comm is not None:
                comm.close()
            raise
        return comm

    def status(self) -> int:
        """Blocking: 0 = fine, bit 0 = some wait timed out (sticky)."""
        err = ctypes.c_uint32()
        check(_lib.lib().chitu_hip_comm_status(self._h, ctypes.byref(err)), "comm_status")
        return err.value

    def set_two_shot(self, min_bytes: int):
        """All-reduces of at least `min_bytes` per rank use the two-shot (reduce-scatter + all-gather) form from now on:
        2 / world of the bytes per link, one more flag hop, bit-identical results.  0 = always; default 256 KB."""
        check(_lib.lib().chitu_hip_comm_set_two_shot(self._h, i64(min_bytes)), "comm_set_two_shot")
        self.two_shot_bytes = int(min_bytes)

    def uses_two_shot(self, rows: int, dim: int) -> bool:
        """The library's choice for an all-reduce of [rows, dim] bf16 (mirrors chitu_hip_comm_allreduce_rmsnorm)."""
        return self.world >= 2 and (dim // 8) % self.world == 0 and rows * dim * 2 >= self.two_shot_bytes

    def poll_error(self) -> int:
        """Non-blocking: the host-visible copy of the error word (0 = nothing reported so far).  Cheap enough to
        be read before every decode step; no stream is synchronised."""
        err = ctypes.c_uint32()
        check(_lib.lib().chitu_hip_comm_poll_error(self._h, ctypes.byref(err)), "comm_poll_error")
        return err.value

    def _between_phases(self):
        """Split-phase mode: this rank's contribution has left (stream drained), every rank's has (host barrier)."""
        import torch.distributed as dist

        torch.cuda.current_stream().synchronize()
        dist.barrier(group=self.split_phase_group)

    def close(self):
        if self._h is not None:
            try:
                closed_buffers.append(self.local_ptr())
            except Exception:  # noqa: BLE001 -- bookkeeping for diagnostics only
                pass
            _lib.lib().chitu_hip_comm_destroy(self._h)
            self._h = None
            _mark_recycled()

    # ------------------------------------------------------------------ collectives
    def fits(self, rows: int, dim: int, terms: int = 1) -> bool:
        return rows <= self.max_rows and dim <= self.max_dim and dim % 8 == 0 and terms <= 16

    def allreduce_rmsnorm(self, part: torch.Tensor, x: Optional[torch.Tensor] = None, weight: Optional[torch.Tensor] = None,
                          eps: float = 1e-6, out_bf16: bool = True, quant: Optional[str] = None, out: Optional[torch.Tensor] = None,
                          phase: int = 0, into=None, tile_major: bool = False):
        """part: this rank's partial, [rows, dim] or [rows, terms, dim] (terms summed first, chitu_hip_moe_sum's
        rounding).  Returns what ops.rms_norm(x, weight, eps, out_bf16, quant, add=<all-reduced part>) returns:
        (x_new, y[, q, s]) -- with x None, x_new is the all-reduced tensor itself; with weight None only
        x_new is returned (a plain all-reduce, `out` may alias `part` for the in-place form).
        phase: 0 = whole collective; 1 = contribute only (returns the output tensors, not yet valid); 2 = complete
        (pass the tuple phase 1 returned as `into`)."""
        if phase == 0 and self.split_phase_group is not None:
            rows_, dim_ = part.shape[0], part.shape[-1]
            first = self.allreduce_rmsnorm(part, x, weight, eps, out_bf16, quant, out, phase=1, tile_major=tile_major)
            self._between_phases()
            if self.uses_two_shot(rows_, dim_):  # the second hop: wait for hop 1, reduce my slice, send it to everyone
                self.allreduce_rmsnorm(part, x, weight, eps, out_bf16, quant, out, phase=3, into=first, tile_major=tile_major)
                self._between_phases()
            return self.allreduce_rmsnorm(part, x, weight, eps, out_bf16, quant, out, phase=2, into=first, tile_major=tile_major)
        require_cuda(part, x, weight)
        assert part.dtype == torch.bfloat16 and part.stride(-1) == 1
        dim = part.shape[-1]
        if part.dim() == 3:
            assert part.is_contiguous()
            rows, terms, term_stride, part_stride = part.shape[0], part.shape[1], dim, part.shape[1] * dim
        else:
            assert part.dim() == 2
            rows, terms, term_stride, part_stride = part.shape[0], 1, 0, part.stride(0)
        if x is not None:
            assert x.dtype == torch.bfloat16 and x.shape == (rows, dim) and x.stride(-1) == 1
        y = q = s = None
        if into is not None:
            sum_out, y, q, s = (tuple(into) + (None,) * 4)[:4] if isinstance(into, tuple) else (into, None, None, None)
            if q is not None and not isinstance(q, torch.Tensor):  # phase 1 handed back an ops.TiledQuant: its buffers
                assert tile_major, "a tile-major result of phase 1 must be completed with tile_major=True"
                q, s = q.q, q.s
            else:
                assert not (tile_major and q is not None), "phase 1 wrote row-major codes; complete it with tile_major=False"
        else:
            sum_out = out if out is not None else torch.empty(rows, dim, dtype=torch.bfloat16, device=part.device)
        assert sum_out.shape == (rows, dim) and sum_out.dtype == torch.bfloat16 and sum_out.stride(-1) == 1
        if weight is not None:
            assert weight.dtype == torch.bfloat16 and weight.is_contiguous() and weight.numel() == dim
            if out_bf16 and y is None:
                y = torch.empty(rows, dim, dtype=torch.bfloat16, device=part.device)
            if quant is not None and q is None:
                if tile_major:  # ops.TiledQuant layout (see chitu_hip_fp8_gemm_blockscale_tm)
                    from .ops import _tiled_buffers

                    q, s = _tiled_buffers(rows, dim, part.device)
                else:
                    q = torch.empty(rows, dim, dtype=torch.float8_e4m3fn, device=part.device)
                    s = torch.empty(rows, dim // 128, dtype=torch.float32, device=part.device)
        else:
            assert quant is None
        check(
            _lib.lib().chitu_hip_comm_allreduce_rmsnorm(
                self._h, ptr(part), i64(part_stride), i32(terms), i64(term_stride), ptr(x),
                i64(x.stride(0) if x is not None else 0), ptr(sum_out), i64(sum_out.stride(0)), ptr(weight), ptr(y), i64(dim),
                i64(rows), i32(dim), f32(eps), ptr(q), ptr(s), i32(_QUANT_MODE[quant] + (4 if tile_major and quant else 0)),
                f32(1e-10), i32(phase), stream_ptr()),
            "comm_allreduce_rmsnorm")
        if weight is None:
            return sum_out
        if quant is not None and tile_major:
            from .ops import TiledQuant

            return (sum_out, y, TiledQuant(q, s, rows, dim), None)
        return (sum_out, y) if quant is None else (sum_out, y, q, s)

    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks of a bf16 tensor [..., dim] (rows <= max_rows)."""
        t2 = t.view(-1, t.shape[-1])
        self.allreduce_rmsnorm(t2, out=t2)
        return t

    def gather_fits(self, rows: int, cols: int) -> bool:
        return cols % 8 == 0 and rows * cols * 2 <= self.gather_bytes

    def all_gather_last_dim(self, y: torch.Tensor, out_dtype=torch.bfloat16, phase: int = 0, into=None) -> torch.Tensor:
        """[rows, cols] bf16 per rank -> [rows, world * cols] (rank-major along the last dim) in bf16 or f32.
        phase / into: as allreduce_rmsnorm."""
        if phase == 0 and self.split_phase_group is not None:
            first = self.all_gather_last_dim(y, out_dtype, phase=1)
            self._between_phases()
            return self.all_gather_last_dim(y, out_dtype, phase=2, into=first)
        require_cuda(y)
        assert y.dim() == 2 and y.dtype == torch.bfloat16 and y.stride(-1) == 1
        rows, cols = y.shape
        out = into if into is not None else torch.empty(rows, self.world * cols, dtype=out_dtype, device=y.device)
        check(_lib.lib().chitu_hip_comm_all_gather(self._h, ptr(y), i64(y.stride(0)), i64(rows), i64(cols), ptr(out),
                                                   i32({torch.bfloat16: 0, torch.float32: 2}[out.dtype]), i32(phase), stream_ptr()),
              "comm_all_gather")
        return out
