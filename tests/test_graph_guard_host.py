"""Host logic of chitu_amd.graphs.capture_verified, on CPU: the capture and the device calls are replaced by
stand-ins, the control flow is the product's -- check by one replay, sweep + repeat on a mismatch, refuse after the last
attempt, and under tensor parallelism the ranks' votes (one rank's mismatch repeats the capture on every rank)."""

import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeGraph:
    def __init__(self, static_out, value, collective=False):
        self.static_out, self.value, self.replays, self.collective = static_out, value, 0, collective

    def replay(self):
        self.replays += 1
        if self.collective:  # a real decode graph holds the step's collectives: every rank must replay the same number of times
            t = torch.ones(1)
            dist.all_reduce(t)
            assert int(t.item()) == dist.get_world_size()
        self.static_out.copy_(self.value)


def _patch(monkeypatch, outcomes, sweeps):
    """graphs._capture hands out fake graphs whose replay writes eager * outcomes[i] (1.0 = a faithful graph)."""
    from chitu_amd import graphs

    made = []

    def fake_capture(step, mode, pool):
        eager = torch.arange(12, dtype=torch.float32).view(3, 4)
        g = _FakeGraph(fake_capture.static_out, eager * outcomes[len(made)])
        made.append(g)
        return g, ("pool", len(made))

    monkeypatch.setattr(graphs, "_capture", fake_capture)
    monkeypatch.setattr(graphs, "sweep_l2", lambda *a, **k: sweeps.append(1))
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    return fake_capture, made


def _run(graphs, fake_capture):
    out = torch.zeros(3, 4)
    fake_capture.static_out = out
    return graphs.capture_verified(lambda: torch.arange(12, dtype=torch.float32).view(3, 4), out, "full", None, what="host test")


def test_a_faithful_capture_is_accepted_after_one_sweep_and_one_replay(monkeypatch):
    from chitu_amd import graphs

    sweeps = []
    fc, made = _patch(monkeypatch, [1.0], sweeps)
    n0 = len(graphs.capture_log)
    g, pool, out = _run(graphs, fc)
    assert g is made[0] and g.replays == 1 and pool == ("pool", 1) and len(sweeps) == 1
    assert graphs.capture_log[n0]["attempts"] == 1 and not graphs.capture_log[n0]["mismatches"]
    del graphs.capture_log[n0:]


def test_a_mismatching_capture_is_swept_and_repeated_and_the_probe_sees_the_failing_state(monkeypatch):
    from chitu_amd import graphs

    sweeps, seen = [], []
    fc, made = _patch(monkeypatch, [1.5, 1.0], sweeps)
    monkeypatch.setattr(graphs, "on_mismatch", lambda ctx: seen.append((ctx["graph"], ctx["info"]["rows_equal"], ctx["mode"])))
    n0 = len(graphs.capture_log)
    g, pool, out = _run(graphs, fc)
    rec = graphs.capture_log[n0]
    assert g is made[1] and rec["attempts"] == 2 and len(rec["mismatches"]) == 1
    m = rec["mismatches"][0]
    assert m["rows_equal"] == 0 and m["rows"] == 3 and m["second_replay_equals_first"] and not m["second_replay_equals_eager"]
    assert seen == [(made[0], 0, "full")] and made[0].replays == 2  # checked, replayed once more for the record
    assert len(sweeps) == 2  # in front of the first capture, and before the repeated one
    assert graphs.unverified_or_retried()[-1] is rec
    del graphs.capture_log[n0:]


def test_a_step_that_never_replays_right_is_refused(monkeypatch):
    from chitu_amd import graphs

    fc, made = _patch(monkeypatch, [2.0, 2.0, 2.0, 2.0], [])
    monkeypatch.setattr(graphs, "on_mismatch", None)
    n0 = len(graphs.capture_log)
    with pytest.raises(RuntimeError, match="does not reproduce the eager step"):
        _run(graphs, fc)
    assert len(made) == graphs._MAX_CAPTURE_ATTEMPTS and graphs.capture_log[n0]["attempts"] == graphs._MAX_CAPTURE_ATTEMPTS
    del graphs.capture_log[n0:]


def _vote_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from chitu_amd import graphs
        from chitu_amd import tensor_parallel as tp

        tp.init_tp(world, 1)
        made = []

        def fake_capture(step, mode, pool):
            eager = torch.arange(12, dtype=torch.float32).view(3, 4)
            # rank 1's FIRST graph is wrong; everything else is faithful
            bad = rank == 1 and len(made) == 0
            g = _FakeGraph(fake_capture.static_out, eager * (3.0 if bad else 1.0), collective=True)
            made.append(g)
            return g, None

        graphs._capture = fake_capture
        graphs.sweep_l2 = lambda *a, **k: None
        torch.cuda.synchronize = lambda *a, **k: None
        out = torch.zeros(3, 4)
        fake_capture.static_out = out
        g, _, _ = graphs.capture_verified(lambda: torch.arange(12, dtype=torch.float32).view(3, 4), out, "full", None, what="vote")
        rec = graphs.capture_log[-1]
        # both ranks captured twice and replayed every graph exactly ONCE: the replays hold the step's collectives (the fake
        # graph all-reduces), so the diagnostic second replay of a failing rank -- which round 4 took on that rank only --
        # would leave it one collective ahead of its peer and hang this test.  Rank 1 has the mismatch on record, rank 0
        # that it was outvoted; both count as a repeated capture.
        assert len(made) == 2 and g is made[1] and rec["attempts"] == 2, (rank, len(made), rec)
        assert [m.replays for m in made] == [1, 1], (rank, [m.replays for m in made])
        assert len(rec["mismatches"]) == 1 and bool(rec["mismatches"][0].get("outvoted")) == (rank == 0), (rank, rec)
        assert "second_replay_equals_first" not in rec["mismatches"][0]
        assert graphs.unverified_or_retried()[-1] is rec
        dist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, "fail: " + repr(e) + traceback.format_exc()))


def test_one_ranks_mismatch_repeats_the_capture_on_every_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_vote_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=30)
    for r in results:
        assert r[1] == "ok", r


def test_eager_canary_sweeps_once_after_a_communicator_event(monkeypatch):
    """graphs.mark_memory_recycled (called by XgmiComm on create / open_peer / close) arms ONE L2 sweep in front of the next
    eager entry point (model.prefill calls sweep_if_memory_was_recycled); nothing is swept otherwise."""
    from chitu_amd import graphs

    sweeps = []
    monkeypatch.setattr(graphs, "sweep_l2", lambda *a, **k: sweeps.append(1))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(graphs, "_recycled_memory_pending", False)
    assert graphs.sweep_if_memory_was_recycled() is False and sweeps == []
    graphs.mark_memory_recycled()
    graphs.mark_memory_recycled()
    assert graphs.sweep_if_memory_was_recycled() is True and sweeps == [1]
    assert graphs.sweep_if_memory_was_recycled() is False and sweeps == [1]
