"""Piecewise hipGraph capture of a decode step that contains collectives.

Reference: chitu/models/model.py:538-622 captures the whole step, NCCL all-reduces included, in one CUDA
graph.  The default here at N > 1 is the same one-graph form on the in-graph xGMI collectives
(tensor_parallel.enable_xgmi).  This module is the FALLBACK for a rank whose collectives go through the
library (RCCL / gloo): the step is replayed as hipGraph PIECES cut at every library collective
(`tensor_parallel._graph_break`) with the collectives issued eagerly between the pieces, so that path never
depends on RCCL calls being capturable; it costs 2 x layers + 2 extra host calls per step and is
bit-identical to eager.
"""

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
