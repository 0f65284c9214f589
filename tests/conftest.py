import os
import sys

import pytest

# before HIP initialises: more hardware queues for the process's streams (test_gpu_xgmi.py runs several ranks of one
# process on their own streams, and kernels that wait for each other must not share a queue); the default is 4
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu via gpurun)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---------------------------------------------------------------- hipGraph captures that failed their replay check
# chitu_amd.graphs.capture_verified rejects a captured decode step whose first replay differs from the eager step and
# captures again.  In the GPU suite every such event is examined IN THE FAILING STATE (round 3's GPUTEST saw a Llama
# graph like that, only inside a whole-suite process) and listed at the end of the run, whether or not a test failed.
def _graph_mismatch_probe(ctx):
    """What can be learned from a rejected graph before it is destroyed; everything goes to stderr and into the
    capture record (chitu_amd.graphs.capture_log).  The steps are ordered from "touches nothing" to "streams 4 GB
    through the chip" and after each one the REJECTED graph is replayed again: the first step after which it reproduces
    the eager logits is what cured it (round 4, first observation: the same graph object replays correctly once 4 GB of
    fills have gone through the device, and keeps doing so)."""
    import time

    import torch

    from chitu_amd import graphs, ops, workspace
    from chitu_amd.attn_backend import HipAttnBackend

    g, static_out, reference, info = ctx["graph"], ctx["static_out"], ctx["reference"], ctx["info"]
    out = {"steps": []}

    def replay_equals_eager(tag):
        static_out.zero_()
        g.replay()
        torch.cuda.synchronize()
        ok = bool(torch.equal(static_out, reference))
        frac = float((static_out != reference).float().mean())
        out["steps"].append((tag, ok, round(frac, 4)))
        return ok

    try:
        snap = torch.cuda.memory_snapshot()
        private = [(s["address"], s["address"] + s["total_size"]) for s in snap if tuple(s.get("segment_pool_id", (0, 0))) != (0, 0)]
        default = [(s["address"], s["address"] + s["total_size"]) for s in snap if tuple(s.get("segment_pool_id", (0, 0))) == (0, 0)]
        out["private_segments"] = [(hex(a), b - a) for a, b in private]
        out["default_segments"] = len(default)
        out["overlapping_segments"] = [(a, b) for a in private for b in default if a[0] < b[1] and b[0] < a[1]][:4]
        out["workspaces_in_private_pools"] = [str(k) for k, t in workspace._ws.items() if any(a <= t.data_ptr() < b for a, b in private)]
        try:  # uncached / IPC buffers this process has freed (xGMI collectives of earlier tests): same addresses?
            from chitu_amd import xgmi

            freed = list(getattr(xgmi, "closed_buffers", []))
            out["freed_uncached_buffers"] = len(freed)
            out["private_segments_on_freed_uncached_addresses"] = [
                (hex(p), hex(a)) for p in freed for a, b in private if a - (4 << 20) <= p < b][:8]
        except Exception as exc:  # noqa: BLE001
            out["freed_uncached_buffers"] = repr(exc)
        replay_equals_eager("rejected graph, third replay, nothing done")
        # the eager step, now, before anything else: still the reference?
        again = ctx["run_eager"]()
        torch.cuda.synchronize()
        out["eager_again_equals_reference"] = bool(torch.equal(again, reference))
        replay_equals_eager("after one more eager step")
        # a fresh capture (new graph object, new private pool), with every op's output kept: does IT replay right, now?
        names = [n for n in ("embed_rope_gather", "bf16_linear", "bf16_linear_add_norm_qkv_post", "bf16_linear_silu_add_norm",
                             "rms_norm", "bf16_linear_add_norm", "bf16_linear_silu", "gqa_qkv_post") if hasattr(ops, n)]
        rec, reals = [None], {}

        def wrap(mod, name):
            real = getattr(mod, name)
            reals[(mod, name)] = real

            def f(*a, **k):
                o = real(*a, **k)
                if rec[0] is not None:
                    outs = o if isinstance(o, (tuple, list)) else (o,)
                    rec[0].append((name, [t.clone() for t in outs if isinstance(t, torch.Tensor)]))
                return o

            setattr(mod, name, f)

        for n in names:
            wrap(ops, n)
        wrap(HipAttnBackend, "attn_with_kvcache")
        try:
            rec[0] = eag = []
            ctx["run_eager"]()
            rec[0] = cap = []
            g2 = torch.cuda.CUDAGraph()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):  # capture_begin / capture_end directly: no empty_cache, no gc in between
                g2.capture_begin()
                out2 = ctx["run_eager"]().clone()
                g2.capture_end()
            torch.cuda.current_stream().wait_stream(side)
            rec[0] = None
            torch.cuda.synchronize()
            g2.replay()
            torch.cuda.synchronize()
            out["fresh_capture_before_any_cure_equals_eager"] = bool(torch.equal(out2, reference))
            diffs = []
            for i, ((n1, a), (n2, b)) in enumerate(zip(eag, cap)):
                for x, y in zip(a, b):
                    if not torch.equal(x.view(torch.uint8), y.view(torch.uint8)):
                        diffs.append((i, n1, tuple(x.shape), round(float((x != y).float().mean()), 4),
                                      bool((y.float() == 0).all()), bool(torch.isnan(y.float()).any())))
            out["fresh_capture_ops"] = len(cap)
            out["fresh_capture_first_differing_ops"] = diffs[:6]
            del g2
        finally:
            rec[0] = None
            for (mod, name), real in reals.items():
                setattr(mod, name, real)
        replay_equals_eager("after the fresh capture")
        time.sleep(0.05)
        replay_equals_eager("after 50 ms of sleep")
        junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")  # a device allocation, not touched
        torch.cuda.synchronize()
        replay_equals_eager("after a 1 GB allocation (untouched)")
        junk[: 64 << 20].fill_(1)
        replay_equals_eager("after a 64 MB fill")
        junk[: 512 << 20].fill_(2)
        replay_equals_eager("after a 512 MB fill")
        for v in range(4):
            junk.fill_(v)
        replay_equals_eager("after 4 GB of fills")
        del junk
        cured = [t for t, ok, _ in out["steps"] if ok]
        out["first_step_that_replayed_right"] = cured[0] if cured else None
        logs = ctx.get("launch_logs")
        if logs is not None:
            e, c = logs
            out["launches_eager"], out["launches_capture"] = len(e), len(c)
            odd = []
            for i, ((n1, a1), (n2, a2)) in enumerate(zip(e, c)):
                if n1 != n2 or len(a1) != len(a2):
                    odd.append((i, n1, n2, "different entry / arity"))
                    continue
                for j, (x, y) in enumerate(zip(a1[:-1], a2[:-1])):  # the last argument is the stream
                    if x != y:
                        in_private = isinstance(y, int) and any(lo <= y < hi for lo, hi in private)
                        in_default = isinstance(x, int) and any(lo <= x < hi for lo, hi in default)
                        if not (in_private and in_default):
                            odd.append((i, n1, j, x, y, "differs, not (default pool -> private pool)"))
            out["launch_argument_anomalies"] = odd[:12]
        out["mem_allocated_MB"] = round(torch.cuda.memory_allocated() / 2**20, 1)
        out["mem_reserved_MB"] = round(torch.cuda.memory_reserved() / 2**20, 1)
        out["graphs_captured_so_far"] = len(graphs.capture_log)
    except Exception as exc:  # noqa: BLE001 -- a diagnostic must not become the failure
        import traceback

        out["probe_error"] = repr(exc) + " | " + traceback.format_exc()[-600:]
    info["probe"] = out
    print(f"[graph mismatch probe] {out}", file=sys.stderr, flush=True)


def pytest_sessionstart(session):
    import torch

    if torch.cuda.is_available():
        from chitu_amd import graphs

        graphs.on_mismatch = _graph_mismatch_probe


def pytest_terminal_summary(terminalreporter):
    try:
        from chitu_amd import graphs
    except Exception:  # noqa: BLE001
        return
    terminalreporter.write_line(f"hipGraph captures checked by one replay against the eager step: {len(graphs.capture_log)}")
    for r in graphs.unverified_or_retried():
        terminalreporter.write_line(f"GRAPH CAPTURE REJECTED AND REPEATED: {r}")
    out = os.path.join(ROOT, "gpurun_out")
    if graphs.capture_log and os.path.isdir(out):
        import json

        with open(os.path.join(out, "graph_capture_log.json"), "w") as f:
            json.dump({"captures": len(graphs.capture_log), "rejected": graphs.unverified_or_retried()}, f, default=str, indent=1)
