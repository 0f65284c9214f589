"""Piecewise hipGraph capture of a decode step that contains collectives.

Reference: chitu/models/model.py:538-622 captures the whole step, NCCL all-reduces included, in one CUDA
graph.  The default here at N > 1 is the same one-graph form on the in-graph xGMI collectives
(tensor_parallel.enable_xgmi).  This module is the FALLBACK for a rank whose collectives go through the
library (RCCL / gloo): the step is replayed as hipGraph PIECES cut at every library collective
(`tensor_parallel._graph_break`) with the collectives issued eagerly between the pieces, so that path never
depends on RCCL calls being capturable; it costs 2 x layers + 2 extra host calls per step and is
bit-identical to eager.
"""

import os
import sys

import torch

from . import tensor_parallel as tp


class PiecewiseGraph:
    """The pieces of one step with the collectives that sit between them (closures over the pieces' static
    tensors, issued on the current stream after the piece that produces their input)."""

    def __init__(self):
        self.pieces = []

    def add(self, graph, run_after):
        self.pieces.append((graph, run_after))

    def replay(self):
        for graph, run_after in self.pieces:
            graph.replay()
            if run_after is not None:
                run_after()


def capture_piecewise(step, pool) -> PiecewiseGraph:
    """Capture `step()` (one eager step on static inputs that writes static outputs; it must have run once
    eagerly already) into pieces that share the memory pool `pool` (torch.cuda.graph_pool_handle())."""
    pieces = PiecewiseGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    cur = [None]

    def begin():
        cur[0] = torch.cuda.CUDAGraph()
        cur[0].capture_begin(pool=pool, capture_error_mode="thread_local")

    def cut(run):  # tensor_parallel._graph_break: a collective sits here
        cur[0].capture_end()
        pieces.add(cur[0], run)
        begin()

    with torch.cuda.stream(side):
        begin()
        tp._graph_break = cut
        try:
            step()
        except BaseException:
            # leave no capture open behind a failing step (the stream would stay in capture mode for good)
            try:
                cur[0].capture_end()
            except Exception:  # noqa: BLE001 -- the original error is the one to report
                pass
            raise
        finally:
            tp._graph_break = None
        cur[0].capture_end()
        pieces.add(cur[0], None)
    torch.cuda.current_stream().wait_stream(side)
    return pieces


def graph_mode(use_graph) -> str:
    """decode(use_graph=...) -> "full" | "piecewise".  A string forces the mode.  True: ONE graph when the rank has
    no library collective in its step -- a single rank, or tensor parallelism on the in-graph xGMI collectives
    (tensor_parallel.enable_xgmi) -- else piecewise, unless CHITU_TP_GRAPH=full asks for the library's calls to
    be captured too.  CHITU_GRAPH_MODE=full|piecewise is a measurement knob that forces a mode on any rank count."""
    import os

    if isinstance(use_graph, str):
        assert use_graph in ("full", "piecewise")
        return use_graph
    if os.environ.get("CHITU_GRAPH_MODE") in ("full", "piecewise"):
        return os.environ["CHITU_GRAPH_MODE"]
    if tp.get_tp_size() == 1 or tp.xgmi_comm() is not None:
        return "full"
    return "full" if os.environ.get("CHITU_TP_GRAPH", "piecewise") == "full" else "piecewise"


# ---------------------------------------------------------------- a captured step is checked before it is trusted
# Every capture of a decode step is followed by ONE replay on the very inputs the eager step in front of the capture
# ran on (the static token / length / block-table buffers; the step's only side effect, the KV row at position L, is
# rewritten with the same bytes), and the replayed logits must equal the eager ones bit for bit -- every launch of the
# step is deterministic.  A graph that fails the check is never returned: the capture is repeated (a fresh graph object)
# and a second failure raises.  Round 3's GPU suite saw a Llama decode graph whose 64 replays all differed from the
# eager launches of the same model in the same process ("graph replay"); whatever produces such a
# graph, it is caught here, at capture time, instead of in the tokens.
capture_log = []      # one record per capture_verified call: {"what", "mode", "attempts", "mismatches": [...]}
on_mismatch = None    # tests: callable(context dict) run in the failing state, before the capture is repeated
_MAX_CAPTURE_ATTEMPTS = int(os.environ.get("CHITU_GRAPH_CAPTURE_ATTEMPTS", "3"))
_SWEEP_BEFORE_CAPTURE = os.environ.get("CHITU_GRAPH_SWEEP_L2", "1") != "0"  # 0: tests of the detection path itself


def _ranks_agree(ok: bool) -> bool:
    """MIN of the ranks' verdicts over the TP group (every rank captures at the same step and must repeat or accept the
    capture together: a verification replay contains the step's collectives)."""
    import torch.distributed as dist

    if tp.get_tp_size() <= 1 or not (dist.is_available() and dist.is_initialized()):
        return ok
    group = tp.get_tp_group()
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    verdict = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(verdict, op=dist.ReduceOp.MIN, group=group)
    return int(verdict.item()) == 1


def _capture(step, mode, pool):
    if mode == "full":
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool):
            step()
        return g, (g.pool() if pool is None else pool)
    if pool is None:
        pool = torch.cuda.graph_pool_handle()
    return capture_piecewise(step, pool), pool


def _describe_mismatch(got, want):
    g, w = got.float(), want.float()
    rows_same = int((got == want).all(-1).sum()) if got.dim() >= 2 else int(torch.equal(got, want))
    return {"rows_equal": rows_same, "rows": int(got.shape[0]) if got.dim() >= 2 else 1,
            "max_abs_diff": float((g - w).abs().max()), "peak": float(w.abs().max()),
            "nan_in_replay": bool(torch.isnan(g).any()), "all_zero_replay": bool((g == 0).all())}


def capture_verified(run_eager, static_out, mode, pool, what="decode step"):
    """Capture `static_out.copy_(run_eager())` in graph mode `mode` ("full" | "piecewise") and return (graph, pool,
    static_out) only once one replay has reproduced, bit for bit, the eager step run just before the capture on the
    same static inputs.  `run_eager()` -> the step's output tensor (logits); static_out None = allocate it.  Raises
    RuntimeError after _MAX_CAPTURE_ATTEMPTS captures that all fail the check."""
    record = {"what": what, "mode": mode, "attempts": 0, "mismatches": []}
    capture_log.append(record)
    # Prevention (the check below is the detection): a capture's private pool is served FRESH device memory, and fresh
    # memory can carry stale L2 lines of its previous owner that survive the kernel-boundary invalidates (round 4: pages
    # recycled from the uncached / IPC-shared exchange buffers of earlier xGMI communicators; one XCD read old rows).  256 MB
    # of ordinary traffic evicts every L2 (8 x 4 MB); ~0.1 ms, once per captured batch size.
    if _SWEEP_BEFORE_CAPTURE:
        sweep_l2()
    for attempt in range(_MAX_CAPTURE_ATTEMPTS):
        record["attempts"] = attempt + 1
        logs = None
        if on_mismatch is not None:  # diagnostic mode: the launches of the eager step and of the capture, argument by argument
            from . import _lib

            logs, _lib.call_log = ([], []), None
        try:
            if logs is not None:
                _lib.call_log = logs[0]
            reference = run_eager().clone()  # also creates every workspace the step needs (workspace.get refuses under capture)
            if static_out is None:
                static_out = torch.zeros_like(reference)
            torch.cuda.synchronize()
            if logs is not None:
                _lib.call_log = logs[1]
            g, new_pool = _capture(lambda: static_out.copy_(run_eager()), mode, pool)
        finally:
            if logs is not None:
                _lib.call_log = None
        torch.cuda.synchronize()  # device-wide, every stream: nothing is in flight when the first replay starts
        static_out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ok = torch.equal(static_out, reference)
        if _ranks_agree(ok):
            return g, new_pool, static_out
        # The vote failed somewhere in the TP group.  What follows must be the SAME sequence of device work on every rank:
        # a replay of the step contains the step's collectives (in-graph xGMI, or RCCL between the pieces), so a diagnostic
        # replay taken only by the rank(s) whose own check failed would leave them one collective step ahead of their peers
        # (xGMI: time-out and error word; RCCL: mismatched collectives, a hang).  Under tensor parallelism the diagnostic
        # replays (and the on_mismatch hook, which replays again) are therefore skipped on every rank; one rank keeps them.
        solo = tp.get_tp_size() <= 1
        if not ok:
            info = _describe_mismatch(static_out, reference)
            info["attempt"] = attempt
            if solo:
                # does the same graph object give the same wrong answer again?  (a property of the graph vs a one-off)
                again = static_out.clone()
                g.replay()
                torch.cuda.synchronize()
                info["second_replay_equals_first"] = bool(torch.equal(static_out, again))
                info["second_replay_equals_eager"] = bool(torch.equal(static_out, reference))
            record["mismatches"].append(info)
            print(f"[chitu_amd] hipGraph capture of {what} ({mode}) failed its replay check, attempt {attempt + 1}: {info}",
                  file=sys.stderr, flush=True)
            if on_mismatch is not None and solo:
                on_mismatch({"graph": g, "run_eager": run_eager, "static_out": static_out, "reference": reference,
                             "mode": mode, "pool": new_pool, "info": info, "launch_logs": logs})
        else:
            record["mismatches"].append({"attempt": attempt, "outvoted": True})  # this rank's replay was right, a peer's was not
        # Round 4's finding: such a replay read stale L2 lines on one XCD -- the capture's private pool had been served
        # memory that was an uncached allocation earlier in the process.  The graph object itself was fine and a NEW
        # capture read the same stale lines; ordinary traffic larger than all L2s cured both for good.  So before the
        # capture is repeated every L2 is swept -- on every rank (local work, no collective).
        sweep_l2()
        del g
        # the rejected graph's pool is not reused: the next attempt allocates its intermediates elsewhere
        pool = None
    raise RuntimeError(f"hipGraph replay of {what} does not reproduce the eager step after {_MAX_CAPTURE_ATTEMPTS} captures; "
                       f"refusing to decode through it: {record['mismatches']}")


_sweep_buffers = {}  # per device; allocated once and kept: a sweep must not become the allocation that runs a full HBM out of memory


def sweep_l2(nbytes: int = 256 << 20):
    """Ordinary write traffic larger than every L2 of the device (8 x 4 MB): evicts whatever lines they hold.  The buffer is
    allocated on first use and kept for the life of the process; if even that allocation fails (KV cache sized to fill the
    HBM) the sweep falls back to smaller buffers written several times, and is skipped -- loudly -- below 32 MB."""
    dev = torch.cuda.current_device()  # one buffer per device: a sweep evicts the L2s of the device it runs on
    buf = _sweep_buffers.get(dev)
    if buf is None or buf.numel() < nbytes:
        size = nbytes
        oom = getattr(torch, "OutOfMemoryError", torch.cuda.OutOfMemoryError)  # (torch < 2.5 has only the cuda-namespaced class)
        while True:
            try:
                with torch.inference_mode(False):  # a normal tensor: it is filled from inference-mode and ordinary callers alike
                    buf = _sweep_buffers[dev] = torch.empty(size, dtype=torch.uint8, device=f"cuda:{dev}")
                break
            except oom:
                size //= 2
                if size < (32 << 20):
                    print("[chitu_amd] sweep_l2: no memory for a sweep buffer; L2 sweep skipped", file=sys.stderr, flush=True)
                    return
    for _ in range(max(1, nbytes // buf.numel())):
        buf.fill_(0)
    torch.cuda.synchronize()


# ---- the eager path's canary (round 5).  capture_verified protects graph captures; eager launches (prefill, eager decode)
# that receive recycled memory have no replay to compare with.  The one known source of stale L2 lines is the life cycle of an
# xGMI communicator's uncached, IPC-shared exchange buffer (DESIGN, "graph replay"): XgmiComm marks the process dirty when such
# a buffer is created, mapped from a peer, or torn down, and the model entry points that launch eagerly (prefill) sweep every L2
# once before their first launch after such an event.  ~0.1 ms, only ever after a communicator event.
_recycled_memory_pending = False


def mark_memory_recycled():
    global _recycled_memory_pending
    _recycled_memory_pending = True


def sweep_if_memory_was_recycled() -> bool:
    """Called by eager entry points before their first launch; True if a sweep ran."""
    global _recycled_memory_pending
    if not _recycled_memory_pending or not torch.cuda.is_available():
        return False
    _recycled_memory_pending = False
    sweep_l2()
    return True


def unverified_or_retried():
    """Captures of this process that needed more than one attempt or were outvoted by a peer rank (bench.py voids its
    line on any, after a MIN over the ranks; tests assert [])."""
    return [r for r in capture_log if r["attempts"] > 1 or r["mismatches"]]
