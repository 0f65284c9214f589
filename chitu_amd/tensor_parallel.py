"""Tensor-parallel groups and linears with the reference's `chitu/tensor_parallel.py` surface.

Reference (read-only): chitu/tensor_parallel.py:1-208.  One process per GPU; the process group
backend string "nccl" IS RCCL on ROCm (collectives run over xGMI and are hipGraph-capturable),
"gloo" is used by the CPU tests.  `linear_op` stays the GEMM plug point (tensor_parallel.py:50,
:125): the DeepSeek modules pass the HIP fp8 linear through it.

All collectives go through `all_reduce` / `defer_all_reduce` / `all_gather_last_dim` below, so the
decode step has ONE place where communication is issued.  Two transports sit behind that seam: the
library (`torch.distributed`: RCCL, or gloo in the CPU tests) and, after `enable_xgmi()`, the in-graph
xGMI kernels of csrc/comm.hip (chitu_amd/xgmi.py) for decode-sized bf16 tensors -- with those the whole
step is one hipGraph, as in the reference (chitu/models/model.py:554-617), and an all-reduce is fused
with the top-k sum in front of it and the residual add + RMSNorm + fp8 quant behind it.
"""

__all__ = [
    "init_tp",
    "get_tp_group",
    "get_tp_size",
    "get_tp_rank",
    "ColumnParallelLinear",
    "RowParallelLinear",
    "VocabParallelEmbedding",
    "enable_xgmi",
    "check_comm",
    "all_reduce",
    "all_gather_last_dim",
]

import os

import torch
import torch.distributed as dist

tp_comm_group = None


def generate_tp_rank_list(tp_size: int, pp_size: int):
    return torch.arange(tp_size * pp_size).reshape(pp_size, tp_size).tolist()


def init_tp(tp_size: int, pp_size: int):
    """Create the TP groups (ranks laid out [pp, tp], tensor_parallel.py:16-27)."""
    global tp_comm_group
    rank_list = generate_tp_rank_list(tp_size, pp_size)
    global_rank = dist.get_rank()
    for ranks in rank_list:
        group = dist.new_group(ranks)
        if global_rank in ranks:
            tp_comm_group = group


def reset_tp():
    global tp_comm_group
    tp_comm_group = None


def get_tp_group():
    return tp_comm_group


def get_tp_size():
    return tp_comm_group.size() if tp_comm_group is not None else 1


def get_tp_rank():
    return dist.get_rank(group=get_tp_group()) if tp_comm_group is not None else 0


# Piecewise hipGraph capture (deepseek_v3.DeepSeekV3Decoder.decode, mode "piecewise"): while a step is
# being captured this is a callable `cut(run)`; every LIBRARY collective below then ENDS the graph piece being
# recorded, hands its own launch (`run`, or None when the group has one rank) to the capture to be issued
# eagerly between the pieces at replay time, and a new piece begins.  No library collective is ever recorded
# into a graph that way, so that path does not depend on RCCL being capturable.  With the in-graph xGMI
# collectives (`enable_xgmi`) there is nothing to cut: they are ordinary kernel launches.
_graph_break = None

# XgmiComm of this rank when the hand-written collectives (csrc/comm.hip) carry the decode step, else None.
_xgmi = None
# What the last enable_xgmi() call decided on this rank and why, and what its pre-flight measured (bench.py copies it into the
# N > 1 line: the first run on real links reports its transport decision and per-transport times even if it then falls back).
xgmi_report = {"enabled": False, "reason": "enable_xgmi was not called"}
PREFLIGHT_BATCHES = (1, 16, 32)


def enable_xgmi(max_rows: int = 64, max_dim: int = 8192, gather_bytes: int = 0, timeout_ms: int = 20000,
                selftest: bool = True) -> bool:
    """Switch the TP group's decode-sized collectives to the in-graph xGMI kernels: every rank creates its
    buffer, the IPC handles travel over the process group, and (selftest) one all-reduce and one all-gather are
    checked against the library's on every rank.  Collective over the TP group, in STAGES (create, map, all-reduce
    self-test in the one-shot and the two-shot form, all-gather self-test): every rank takes part in every collective of a stage whatever happened to it
    locally, the ranks agree on the stage's verdict (MIN over the group), and only a unanimous success moves on --
    so a failure on some ranks (IPC refused on one GPU, a mismatch on one rank) can never leave the others inside a
    collective nobody else enters.  On any failure the library path stays in place on EVERY rank and False is returned."""
    global _xgmi, xgmi_report
    if get_tp_size() <= 1 or not torch.cuda.is_available():
        xgmi_report = {"enabled": False, "stage": "none", "reason": "one rank or no device: nothing to enable"}
        return False
    from .xgmi import XgmiComm

    stage = ["create + map (IPC handles)"]

    group, rank, world = get_tp_group(), get_tp_rank(), get_tp_size()
    lib_dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"  # the library's answer is computed where its backend works

    def unanimous(ok: bool) -> bool:
        verdict = torch.tensor([1 if ok else 0], dtype=torch.int32, device=lib_dev)
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)
        return int(verdict.item()) == 1

    def give_up(comm, why):
        global xgmi_report
        xgmi_report = {"enabled": False, "stage": stage[0], "reason": why or "a peer rank failed this stage (its own line says why)",
                       "fallback": "library collectives (RCCL / gloo) between pieces of the step's graph (graphs.py)"}
        if why:
            print(f"[chitu_amd] rank {rank}: xGMI collectives not enabled ({why}); using the library path", flush=True)
        if comm is not None:
            try:
                comm.close()
            except Exception as e:  # noqa: BLE001 -- a failing clean-up must not take this rank out of the stages its peers still run
                print(f"[chitu_amd] rank {rank}: closing the xGMI buffers failed ({e!r})", flush=True)
        return False

    try:  # stages 1 + 2 (create, map): from_group is collective-safe and raises on every rank or on none
        comm = XgmiComm.from_group(group, max_rows=max_rows, max_dim=max_dim, gather_bytes=gather_bytes, timeout_ms=timeout_ms)
    except Exception as e:  # noqa: BLE001 -- from_group raises only AFTER its agree() round, on every rank alike; whatever the type, the library path stays
        return give_up(None, repr(e))
    if os.environ.get("CHITU_XGMI_SPLIT_PHASE", "0") == "1":
        # every collective as contribute -> host barrier over the group -> complete: no kernel waits for a peer (rank
        # processes time-sliced on ONE GPU: tools/xgmi_world8.py, tests/test_gpu_xgmi.py); eager launches only
        comm.split_phase_group = group
    two_shot_note = "not tested (selftest off)"
    if selftest:
        gen = torch.Generator(device="cuda").manual_seed(1234 + rank)
        dim = min(max_dim, 1024)
        stage[0] = "one-shot all-reduce self-test"
        for rows in (1, min(max_rows, 5)):  # stage 3: the library's collective is entered by every rank, unconditionally
            part = torch.randn(rows, dim, device="cuda", generator=gen).to(torch.bfloat16)
            want = part.float().to(lib_dev)
            dist.all_reduce(want, group=group)  # fp32 sum: order-free up to rounding
            ok, why = True, ""
            try:
                got = comm.allreduce_rmsnorm(part).float().to(lib_dev)
                if comm.status() != 0 or not torch.allclose(got, want, rtol=2e-2, atol=2e-2):
                    ok, why = False, "all-reduce self-test mismatch / timeout"
            except Exception as e:  # noqa: BLE001 -- a local failure is a vote, not an exit
                ok, why = False, repr(e)
            if not unanimous(ok):
                return give_up(comm, why)
        # stage 3b: the two-shot form (reduce-scatter + all-gather inside the launch, the form of >= 256 KB messages) on the
        # same check, twice (both data-slot parities).  A timeout leaves the sticky error word set: give up as above.  A
        # group on which only its VALUES are wrong keeps the xGMI collectives in the one-shot form for every size.
        stage[0] = "two-shot all-reduce self-test"
        two_shot_note = "not applicable (dim / 8 not divisible by the world size)"
        if (dim // 8) % world == 0:
            default_threshold, values_ok = comm.two_shot_bytes, True
            comm.set_two_shot(0)
            for _ in range(2):
                part = torch.randn(min(max_rows, 4), dim, device="cuda", generator=gen).to(torch.bfloat16)
                want = part.float().to(lib_dev)
                dist.all_reduce(want, group=group)
                alive, same, why = True, True, ""
                try:
                    got = comm.allreduce_rmsnorm(part).float().to(lib_dev)
                    alive = comm.status() == 0
                    same = alive and torch.allclose(got, want, rtol=2e-2, atol=2e-2)
                    why = "" if alive else "two-shot all-reduce self-test timeout"
                except Exception as e:  # noqa: BLE001
                    alive, same, why = False, False, repr(e)
                if not unanimous(alive):
                    return give_up(comm, why)
                values_ok = unanimous(same) and values_ok
            comm.set_two_shot(default_threshold if values_ok else 1 << 62)
            two_shot_note = (f"in use from {default_threshold} bytes per rank" if values_ok
                             else "values mismatched in the self-test: one-shot form kept for every size")
            if not values_ok:
                print(f"[chitu_amd] rank {rank}: two-shot all-reduce self-test mismatch; keeping the one-shot form for every size", flush=True)
        stage[0] = "all-gather self-test"
        if gather_bytes >= 2 * 32 * 2:  # stage 4
            y = torch.randn(2, 32, device="cuda", generator=gen).to(torch.bfloat16)
            ref = [torch.empty(2, 32, dtype=torch.float32, device=lib_dev) for _ in range(world)]
            dist.all_gather(ref, y.float().to(lib_dev), group=group)
            ok, why = True, ""
            try:
                got = comm.all_gather_last_dim(y).float().to(lib_dev)
                if comm.status() != 0 or not torch.equal(got, torch.cat(ref, dim=-1)):
                    ok, why = False, "all-gather self-test mismatch / timeout"
            except Exception as e:  # noqa: BLE001
                ok, why = False, repr(e)
            if not unanimous(ok):
                return give_up(comm, why)
    _xgmi = comm
    xgmi_report = {"enabled": True, "stage": "all stages passed" if selftest else "created (self-test skipped)", "reason": "",
                   "two_shot": two_shot_note, "world": world,
                   "mode": "split-phase (host barrier inside every collective: rank processes sharing one GPU)"
                           if comm.split_phase_group is not None else "in-graph (kernels wait for their peers)"}
    if selftest and os.environ.get("CHITU_XGMI_PREFLIGHT", "1") != "0":
        # stage 5, never a reason to fall back: per-transport times at the bench's batch sizes, BEFORE anything is captured --
        # the first run on real links yields the threshold data even if a later capture is rejected.  Collective: every rank runs it.
        try:
            xgmi_report["preflight_us"] = _preflight_times(comm, group, min(max_dim, 7168), gather_bytes)
        except Exception as e:  # noqa: BLE001 -- a report, not a gate (a timeout here shows up in check_comm() like any other)
            xgmi_report["preflight_us"] = f"failed: {e!r}"[:200]
    return True


def _preflight_times(comm, group, dim, gather_bytes, iters=10):
    """Eager us per call (HIP events over `iters` calls, after 2 warm-up calls; every rank issues the same sequence) of the
    all-reduce in its one-shot and two-shot form and of the logits-sized all-gather, at PREFLIGHT_BATCHES rows of `dim`."""
    out = {}
    default = comm.two_shot_bytes
    gen = torch.Generator(device="cuda").manual_seed(77)

    def eager_us(fn):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        dist.barrier(group=group)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) * 1e3 / iters, 2)

    try:
        for b in PREFLIGHT_BATCHES:
            if not comm.fits(b, dim):
                continue
            part = torch.randn(b, dim, device="cuda", generator=gen).to(torch.bfloat16)
            row = {"message_KB_per_rank": round(b * dim * 2 / 1024, 1)}
            comm.set_two_shot(1 << 60)
            row["one_shot_us"] = eager_us(lambda: comm.allreduce_rmsnorm(part))
            if (dim // 8) % comm.world == 0 and default < (1 << 61):
                comm.set_two_shot(0)
                row["two_shot_us"] = eager_us(lambda: comm.allreduce_rmsnorm(part))
            comm.set_two_shot(default)
            row["form_in_step"] = "two-shot" if comm.uses_two_shot(b, dim) else "one-shot"
            cols = gather_bytes // (2 * max(PREFLIGHT_BATCHES)) if gather_bytes else 0
            if cols >= 32 and comm.gather_fits(b, cols):
                y = torch.randn(b, cols, device="cuda", generator=gen).to(torch.bfloat16)
                row["all_gather_us"] = eager_us(lambda: comm.all_gather_last_dim(y))
                row["all_gather_cols_per_rank"] = cols
            out[f"bs{b}"] = row
    finally:
        comm.set_two_shot(default)
    out["status_after"] = int(comm.status())
    return out


class CollectiveTimeout(RuntimeError):
    """An in-graph collective gave up waiting for a peer (csrc/comm.hip: bounded spins, sticky error word)."""


def check_comm():
    """Raise CollectiveTimeout if a collective of an EARLIER step timed out.  The kernels cannot raise: a timed-out wait
    sets the error word, reduces whatever is in the slots and every later collective skips its wait, so the tokens that
    follow are garbage -- the reference, in that situation, has NCCL's watchdog abort the process.  This is the host
    side of that contract: a non-blocking read of the pinned copy of the error word, called by the decoders before
    every step (and, blocking, at the end of generate()), so a stalled peer becomes an exception within one step."""
    if _xgmi is not None and _xgmi.poll_error() != 0:
        raise CollectiveTimeout(
            f"rank {get_tp_rank()}: an xGMI collective timed out waiting for a peer; every result since is invalid")


def disable_xgmi():
    global _xgmi
    if _xgmi is not None:
        _xgmi.close()
    _xgmi = None


def xgmi_comm():
    return _xgmi


def xgmi_split_phase() -> bool:
    """The in-graph collectives are running in their split-phase test form (a host barrier inside every collective):
    the decoders then launch eagerly -- such a step cannot be captured."""
    return _xgmi is not None and _xgmi.split_phase_group is not None


class PendingAllReduce:
    """A rank's partial ([rows, dim] or the fused MoE's un-summed [rows, terms, dim]) whose all-reduce has been
    handed to the kernel that consumes it: [top-k sum ->] all-reduce -> residual add -> RMSNorm -> quant is one
    launch (XgmiComm.allreduce_rmsnorm).  `resolve()` performs the plain all-reduce instead."""

    __slots__ = ("part",)

    def __init__(self, part):
        self.part = part

    def resolve(self) -> torch.Tensor:
        return _xgmi.allreduce_rmsnorm(self.part)


def resolve(t):
    return t.resolve() if isinstance(t, PendingAllReduce) else t


def _xgmi_fits(t: torch.Tensor) -> bool:
    return (_xgmi is not None and t.is_cuda and t.dtype == torch.bfloat16 and t.dim() >= 2 and t.is_contiguous()
            and _xgmi.fits(t.numel() // t.shape[-1], t.shape[-1]))


def all_reduce(t: torch.Tensor) -> torch.Tensor:
    """Sum over the TP group, in place (tensor_parallel.py:166, model_deepseek_v3.py:1011)."""
    if get_tp_size() > 1 and _xgmi_fits(t):
        return _xgmi.all_reduce_(t)
    run = (lambda: dist.all_reduce(t, group=get_tp_group())) if get_tp_size() > 1 else None
    if _graph_break is not None:
        _graph_break(run)
    elif run is not None:
        run()
    return t


def defer_all_reduce(t: torch.Tensor):
    """The all-reduce of a decode-step partial that the next residual-add + RMSNorm consumes.  t: [rows, dim], or
    the fused MoE's un-summed [rows, terms, dim] when `defers_topk_sum()`.  One rank: t itself (the norm sums
    the terms); xGMI collectives: a PendingAllReduce (the norm launch does the reduction); library path: the
    all-reduced tensor."""
    if get_tp_size() == 1:
        if _graph_break is not None:
            _graph_break(None)  # piecewise replay on one rank: the cut stays where a library collective would sit
        return t
    if _xgmi is not None and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous() and (
            (t.dim() == 2 and _xgmi.fits(t.shape[0], t.shape[1])) or
            (t.dim() == 3 and _xgmi.fits(t.shape[0], t.shape[2], t.shape[1]))):
        return PendingAllReduce(t)
    assert t.dim() == 2, "the library all-reduce needs the summed tensor"
    return all_reduce(t)


def add_norm(x, pending, weight, eps, out_bf16=True, quant=None, tile_major=False):
    """(x_new, y, q, s) with x_new = x + pending and y / (q, s) = RMSNorm(x_new) [fp8-quantised]; entries not asked
    for are None.  pending: None | a tensor ([rows, dim], or [rows, terms, dim]: terms summed first) | a
    PendingAllReduce (this rank's partial: all-reduced over xGMI inside the same launch).  The one place where a
    residual add, its norm and -- under tensor parallelism -- the all-reduce in front of them meet.
    tile_major: q comes back as an ops.TiledQuant (s None) wherever a residual is folded in (the wide row form)."""
    from . import ops

    if isinstance(pending, PendingAllReduce) and quant == "int8":
        pending = pending.resolve()  # (the fused all-reduce launch has the fp8 quantisers only)
    if isinstance(pending, PendingAllReduce):
        res = _xgmi.allreduce_rmsnorm(pending.part, x, weight, eps, out_bf16=out_bf16, quant=quant, tile_major=tile_major)
    elif pending is None:  # (no residual yet: the narrow row form, row-major output whatever tile_major asks for)
        r = ops.rms_norm(x, weight, eps, out_bf16=out_bf16, quant=quant)
        res = (x,) + (r if isinstance(r, tuple) else (r,))
    elif tile_major:
        res = ops.rms_norm(x, weight, eps, out_bf16=out_bf16, quant=quant, add=pending, tile_major=True)
    else:
        res = ops.rms_norm(x, weight, eps, out_bf16=out_bf16, quant=quant, add=pending)
    return tuple(res) + (None,) * (4 - len(res))


def defers_topk_sum(rows: int, dim: int, terms: int) -> bool:
    """May the fused MoE leave its top-k sum to the consumer of defer_all_reduce()?  Yes when nothing sits
    between the experts and the next norm (one rank) or when the in-graph all-reduce takes the terms."""
    if terms > 16:  # chitu_hip_rmsnorm / comm_allreduce_rmsnorm sum <= 16 terms
        return False
    return get_tp_size() == 1 or (_xgmi is not None and _xgmi.fits(rows, dim, terms))


def all_gather_last_dim(y: torch.Tensor, out_dtype=None) -> torch.Tensor:
    """Concatenate the last dimension over TP ranks (tensor_parallel.py:94-102: the reference
    gathers on a permuted [N/tp, ...] layout so the result is rank-major along the last dim).  out_dtype:
    cast of the result folded into the gather (the logits' `.float()`)."""
    tp = get_tp_size()
    if tp == 1:
        return y if out_dtype is None else y.to(out_dtype)
    if (_xgmi is not None and y.is_cuda and y.dtype == torch.bfloat16 and y.stride(-1) == 1
            and out_dtype in (None, torch.bfloat16, torch.float32)):
        y2 = y.reshape(-1, y.shape[-1])
        if _xgmi.gather_fits(y2.shape[0], y2.shape[1]):
            out = _xgmi.all_gather_last_dim(y2, out_dtype or torch.bfloat16)
            return out.view(*y.shape[:-1], out.shape[-1])
    y_t = y.permute(-1, *range(y.dim() - 1)).contiguous()
    shape = list(y_t.shape)
    shape[0] *= tp
    gathered = y.new_empty(shape)
    run = lambda: dist.all_gather_into_tensor(gathered, y_t, group=get_tp_group())
    if _graph_break is not None:
        _graph_break(run)
    else:
        run()
    out = gathered.permute(*range(1, y.dim()), 0)
    # no cast: the reference's permuted view onto the gather's buffer (tensor_parallel.py:101-102).  With the cast (the
    # logits' `.float()`): a DENSE result -- a cast keeps its input's strides unless told otherwise, and the sampler reads
    # logits rows with a unit stride
    if out_dtype is None or out_dtype == out.dtype:
        return out
    return out.to(out_dtype, memory_format=torch.contiguous_format)


class _ShardedLinear(torch.nn.Module):
    """Shared part of the two Megatron linears: group bookkeeping and the `[out, in]` weight (and bias) of THIS
    rank, `split` naming the dimension the full matrix is cut along ("out": rows / column-parallel, "in":
    columns / row-parallel).  Parameters are uninitialised, like the reference's (the loader fills them)."""

    def __init__(self, in_features, out_features, split, has_bias, dtype, bias_dtype, linear_op):
        super().__init__()
        self.tp_group, self.tp_size, self.rank = get_tp_group(), get_tp_size(), get_tp_rank()
        self.in_features, self.out_features, self.linear_op = in_features, out_features, linear_op
        cut = out_features if split == "out" else in_features
        assert cut % self.tp_size == 0, f"{split}_features must be divisible by tp_size"
        rows = out_features // self.tp_size if split == "out" else out_features
        cols = in_features if split == "out" else in_features // self.tp_size
        self.weight = torch.nn.Parameter(torch.empty(rows, cols, dtype=dtype), requires_grad=False)
        self.bias = torch.nn.Parameter(torch.empty(rows, dtype=bias_dtype or dtype), requires_grad=False) if has_bias else None


class ColumnParallelLinear(_ShardedLinear):
    """y = x W_r^T (+ b_r) with the OUTPUT features split over the ranks; gather_output concatenates the ranks'
    slices along the last dimension (tensor_parallel.py:42-103)."""

    def __init__(self, in_features, out_features, has_bias=True, gather_output=True, dtype=None, bias_dtype=None,
                 linear_op=torch.nn.functional.linear):
        super().__init__(in_features, out_features, "out", has_bias, dtype, bias_dtype, linear_op)
        self.gather_output = gather_output

    def forward(self, x):
        y = self.linear_op(x, self.weight, self.bias)
        return all_gather_last_dim(y) if self.gather_output and self.tp_size > 1 else y


class RowParallelLinear(_ShardedLinear):
    """y = sum over ranks of x_r W_r^T with the INPUT features split; the bias is added once (by rank 0, before the
    all-reduce); an input that is not already this rank's slice is sliced here (tensor_parallel.py:106-169)."""

    def __init__(self, in_features, out_features, has_bias=True, input_is_parallel=False, dtype=None, bias_dtype=None,
                 linear_op=torch.nn.functional.linear):
        super().__init__(in_features, out_features, "in", has_bias, dtype, bias_dtype, linear_op)
        self.input_is_parallel = input_is_parallel

    def forward(self, x):
        if self.tp_size == 1:
            return self.linear_op(x, self.weight, self.bias)
        if not self.input_is_parallel:
            x = x.unflatten(-1, (self.tp_size, -1)).select(-2, self.rank)
        return all_reduce(self.linear_op(x, self.weight, self.bias if self.rank == 0 else None))


class VocabParallelEmbedding(torch.nn.Module):
    """Embedding table split by vocabulary rows: ids outside this rank's range look up row 0 and are zeroed, the
    all-reduce assembles the batch (tensor_parallel.py:172-208).  The caller's id tensor is not modified."""

    def __init__(self, num_embeddings, embedding_dim, dtype=None):
        super().__init__()
        self.tp_group, self.tp_size, self.rank = get_tp_group(), get_tp_size(), get_tp_rank()
        assert num_embeddings % self.tp_size == 0, "num_embeddings must be divisible by tp_size"
        per_rank = num_embeddings // self.tp_size
        self.vocab_start_idx, self.vocab_end_idx = self.rank * per_rank, (self.rank + 1) * per_rank
        self.weight = torch.nn.Parameter(torch.empty(per_rank, embedding_dim, dtype=dtype), requires_grad=False)

    def forward(self, x):
        if self.tp_size == 1:
            return torch.nn.functional.embedding(x, self.weight)
        local = x - self.vocab_start_idx
        foreign = (local < 0) | (local >= self.weight.shape[0])
        y = torch.nn.functional.embedding(local.masked_fill(foreign, 0), self.weight)
        return all_reduce(y.masked_fill(foreign.unsqueeze(-1), 0))
