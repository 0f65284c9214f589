"""HIP sampler (csrc/sample.hip via chitu_amd.sampling) vs the oracle and the reference-generated
fixture: frequency penalty and greedy bit-exact, the probability-space sampler bit-exact against
the integer specification, the logits-space sampler within the tolerance its exp allows."""

import numpy as np
import pytest
import torch

from oracle import sampling as osmp
from tests.util import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["candidates", "radix"], autouse=True)
def sampler_path(request, monkeypatch):
    """Every test runs twice: with the short candidate-list path enabled (the default; it declines where it
    cannot prove the list complete and the radix descent runs) and with the radix descent alone
    (launch-variant override "sample_radix", chitu_hip_debug_option).  Both implement one integer specification."""
    from chitu_amd._lib import debug_option

    if request.param == "radix":
        with debug_option("sample_radix", 1):
            yield request.param
    else:
        yield request.param


def _responses(g):
    off = g["resp_off"].tolist()
    flat = g["resp_flat"].tolist()
    return [flat[off[i]: off[i + 1]] for i in range(len(off) - 1)]


def _rows(kind, rows, vocab, seed):
    g = torch.Generator().manual_seed(seed)
    if kind == "peaked":
        return torch.randn(rows, vocab, generator=g) * 3.0
    if kind == "flat":
        return torch.randn(rows, vocab, generator=g) * 0.01
    if kind == "ties":
        return torch.round(torch.randn(rows, vocab, generator=g) * 2.0)
    if kind == "equal":
        return torch.zeros(rows, vocab)
    if kind == "underflow":
        return torch.randn(rows, vocab, generator=g) * 40.0
    raise ValueError(kind)


# ---------------------------------------------------------------- frequency penalty
def test_frequency_penalty_reference_fixture():
    from chitu_amd import sampling

    g = golden("sampler")
    lg = torch.from_numpy(g["logits"].copy()).cuda()
    out = sampling.apply_frequency_penalty(lg, _responses(g), g["penalties"].tolist())
    assert out.data_ptr() == lg.data_ptr()  # in place, like index_add_
    assert np.array_equal(out.cpu().numpy().view(np.uint32), g["penalised_logits"].view(np.uint32))


def test_frequency_penalty_ragged_duplicates_and_bad_ids():
    from chitu_amd import sampling

    rng = np.random.default_rng(0)
    rows, vocab = 5, 777
    base = torch.from_numpy(rng.normal(0, 1, (rows, vocab)).astype(np.float32))
    responses = [list(rng.integers(0, vocab, 3000)), [], [4] * 500, list(rng.integers(0, 10, 64)), [776, 0, 776]]
    pens = [0.25, 1.0, 0.125, 0.0, 3.5]
    ref = osmp.frequency_penalty(base.clone(), responses, pens)
    # a row view with a stride wider than the vocabulary, and ids outside [0, vocab) that must be ignored
    wide = torch.zeros(rows, vocab + 19).cuda()
    view = wide[:, 3: 3 + vocab]
    view.copy_(base)
    dev_resp = [r + [-1, vocab, vocab + 5] for r in responses]
    sampling.apply_frequency_penalty(view, dev_resp, pens)
    assert np.array_equal(view.cpu().numpy().view(np.uint32), ref.numpy().view(np.uint32))
    assert float(wide[:, :3].abs().sum()) == 0.0 and float(wide[:, 3 + vocab:].abs().sum()) == 0.0


# ---------------------------------------------------------------- greedy
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("vocab", [1, 3, 4, 5, 1000, 1023, 4099, 129280])
def test_argmax_exact(dtype, vocab):
    from chitu_amd import sampling

    rows = 7
    x = (_rows("peaked", rows, vocab, vocab) if vocab > 5 else _rows("ties", rows, vocab, vocab)).to(dtype)
    x[1] = torch.round(x[1].float()).to(dtype)  # ties: the first maximum wins
    if vocab > 4:
        x[2, vocab - 1] = 100.0  # maximum in the ragged tail
    want = np.argmax(x.float().numpy(), axis=-1)
    got = sampling.argmax(x.cuda()).cpu().numpy()
    assert np.array_equal(got, want)
    # unaligned rows: a column-offset view (16-B vector loads are illegal there)
    wide = torch.zeros(rows, vocab + 7, dtype=dtype)
    wide[:, 1: 1 + vocab] = x
    got = sampling.argmax(wide.cuda()[:, 1: 1 + vocab]).cpu().numpy()
    assert np.array_equal(got, want)


def test_greedy_reference_fixture_and_all_greedy_dispatch():
    from chitu_amd import sampling

    g = golden("sampler")
    lg = torch.from_numpy(g["logits"].copy()).cuda()
    tok = sampling.sample_tokens(lg, g["temperatures"].tolist(), [1] * 8, g["top_ps"].tolist(),
                                 frequency_penalties=g["penalties"].tolist(), responses=_responses(g))
    assert np.array_equal(tok.cpu().numpy(), g["greedy_tokens"])
    # top_k <= 1 everywhere is the reference's is_all_greedy (task.py:457), 0 and -1 included
    tok = sampling.sample_tokens(torch.from_numpy(g["penalised_logits"]).cuda(), g["temperatures"].tolist(),
                                 [1, 0, -1, 1, 1, 0, 1, 1], g["top_ps"].tolist())
    assert np.array_equal(tok.cpu().numpy(), g["greedy_tokens"])


# ---------------------------------------------------------------- sampling: bit-exact in probability space
CASES = [  # (kind, vocab, top_k, top_p)
    ("peaked", 1000, 50, 0.9), ("peaked", 4099, -1, 0.95), ("peaked", 257, 5, 1.0), ("peaked", 5, 3, 0.5),
    ("flat", 1000, 50, 0.9), ("flat", 4099, 0, 0.3), ("flat", 32000, 3000, 0.999), ("ties", 1000, 20, 0.9),
    ("ties", 4099, -1, 0.7), ("equal", 1000, 17, 0.5), ("equal", 257, -1, 1.0), ("underflow", 1000, 50, 0.9),
    ("underflow", 4099, -1, 1.0), ("peaked", 129280, 50, 0.9), ("flat", 129280, -1, 0.9), ("ties", 129280, 40000, 0.6),
    ("peaked", 1000, 1, 0.9), ("peaked", 1000, 2, 0.0),
]


@pytest.mark.parametrize("kind,vocab,top_k,top_p", CASES)
def test_probability_space_sampler_is_bit_exact(kind, vocab, top_k, top_p):
    """probs_mode: every operation of the kernel is an IEEE operation, so token and kept count
    must equal the integer specification's (top_k_top_p_min_p_sampling_from_probs_torch, utils.py:62)."""
    from chitu_amd import sampling

    rows = 6 if vocab <= 4099 else 3
    probs = torch.softmax(_rows(kind, rows, vocab, 7) / 0.8, dim=-1)
    us = [0.0, 0.99999994, 0.5, 0.123456, 0.87654, 0.3333][:rows]
    tok, n_kept, mass = sampling.top_k_top_p_min_p_sampling_from_probs_torch(
        probs.cuda(), [top_k] * rows, [top_p] * rows, uniforms=us, return_stats=True)
    tok, n_kept, mass = tok.cpu().numpy(), n_kept.cpu().numpy(), mass.cpu().numpy()
    for r in range(rows):
        t_o, n_o, m_o = osmp.sample_fixed_point(probs[r].numpy(), 1.0, top_k, top_p, us[r], probs_mode=True)
        assert (int(tok[r]), int(n_kept[r])) == (t_o, n_o), (r, tok[r], n_kept[r], t_o, n_o)
        if top_k != 1:
            assert abs(float(mass[r]) - m_o) < 1e-6


def test_mixed_parameters_per_row_and_strided_input():
    from chitu_amd import sampling

    rows, vocab = 9, 2001  # odd vocab: ragged tail; odd stride: unaligned rows
    probs = torch.softmax(_rows("peaked", rows, vocab, 11), dim=-1)
    ks = [50, 1, -1, 7, 2001, 3000, 2, 0, 100]
    ps = [0.9, 0.5, 0.2, 1.0, 0.99, 0.8, 0.0, 0.6, 1.5]
    us = [0.1 * i + 0.05 for i in range(rows)]
    wide = torch.zeros(rows, vocab + 6).cuda()
    view = wide[:, 5: 5 + vocab]
    view.copy_(probs)
    tok, n_kept, _ = sampling.top_k_top_p_min_p_sampling_from_probs_torch(view, ks, ps, uniforms=us, return_stats=True)
    for r in range(rows):
        t_o, n_o, _ = osmp.sample_fixed_point(probs[r].numpy(), 1.0, ks[r], ps[r], us[r], probs_mode=True)
        assert (int(tok[r]), int(n_kept[r])) == (t_o, n_o), r


# ---------------------------------------------------------------- sampling from logits (the decode step's form)
@pytest.mark.parametrize("kind,vocab,top_k,top_p", CASES[:14])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_logits_sampler_within_exp_tolerance(kind, vocab, top_k, top_p, dtype):
    """exp on the GPU differs from numpy's by an ulp, so the boundary may move by elements of
    negligible mass: the kept count within 2 and the kept mass within 1e-4 of the specification,
    the drawn token either the specification's or one whose CDF interval is within 1e-4 of u."""
    from chitu_amd import sampling

    rows = 6 if vocab <= 4099 else 2
    temps = [0.8, 1.0, 0.5, 1.3, 0.7, 2.0][:rows]
    x = _rows(kind, rows, vocab, 3).to(dtype)
    us = [0.0, 0.99999994, 0.5, 0.123456, 0.87654, 0.3333][:rows]
    tok, n_kept, mass = sampling.top_k_top_p_sampling_from_logits(x.cuda(), temps, [top_k] * rows, [top_p] * rows,
                                                                  uniforms=us, return_stats=True)
    tok, n_kept, mass = tok.cpu().numpy(), n_kept.cpu().numpy(), mass.cpu().numpy()
    exact = 0
    for r in range(rows):
        row = x[r].float().numpy()
        t_o, n_o, m_o = osmp.sample_fixed_point(row, temps[r], top_k, top_p, us[r])
        if top_k == 1:
            assert int(tok[r]) == t_o
            exact += 1
            continue
        assert abs(int(n_kept[r]) - n_o) <= 2, (r, n_kept[r], n_o)
        assert abs(float(mass[r]) - m_o) < 1e-4
        if int(tok[r]) == t_o:
            exact += 1
            continue
        dist = osmp.kept_distribution(row, temps[r], top_k, top_p)
        cdf = np.cumsum(dist)
        t = int(tok[r])
        lo = cdf[t] - dist[t]
        assert lo - 1e-4 <= us[r] <= cdf[t] + 1e-4, (r, t, t_o, lo, cdf[t], us[r])
    assert exact >= rows - 1


def test_logits_sampler_on_reference_fixture():
    """Kept counts equal the reference's own masks (tests/golden/sampler.npz) and every draw lands in
    the reference's kept set."""
    from chitu_amd import sampling

    g = golden("sampler")
    lg = torch.from_numpy(g["penalised_logits"]).cuda()
    masked = g["masked_sorted"]
    probs = torch.from_numpy(g["probs"])
    _, idx = probs.sort(dim=-1, descending=True)
    n_ref = (masked > 0).sum(axis=1)
    for u in (0.0, 0.25, 0.5, 0.75, 0.99999994):
        tok, n_kept, _ = sampling.top_k_top_p_sampling_from_logits(
            lg, g["temperatures"].tolist(), g["top_ks"].tolist(), g["top_ps"].tolist(), uniforms=[u] * 8, return_stats=True)
        tok, n_kept = tok.cpu().numpy(), n_kept.cpu().numpy()
        for row in range(8):
            if int(g["top_ks"][row]) == 1:
                assert tok[row] == g["greedy_tokens"][row]
                continue
            assert abs(int(n_kept[row]) - int(n_ref[row])) <= 1, (row, n_kept[row], n_ref[row])
            kept_ref = set(idx[row, : int(n_ref[row])].tolist())
            if row != 3:  # row 3 has tied probabilities: which tie members survive is unspecified upstream
                assert int(tok[row]) in kept_ref, (row, u)
            else:
                assert float(probs[row, int(tok[row])]) >= float(probs[row, idx[row, int(n_ref[row]) - 1]]) - 1e-9


def test_draws_follow_the_kept_distribution():
    """8192 rows of the same logits with independent uniforms from a CUDA generator: the histogram of
    tokens matches the specification's kept distribution (5-sigma binomial bound per token), and
    nothing outside the kept set is ever drawn."""
    from chitu_amd import sampling

    rows, vocab = 8192, 300
    row = _rows("peaked", 1, vocab, 5) * 0.7
    x = row.expand(rows, vocab).contiguous().cuda()
    gen = torch.Generator(device="cuda").manual_seed(1234)
    tok = sampling.top_k_top_p_sampling_from_logits(x, [0.9] * rows, [12] * rows, [0.95] * rows, generator=gen)
    counts = np.bincount(tok.cpu().numpy(), minlength=vocab).astype(np.float64)
    dist = osmp.kept_distribution(row[0].numpy(), 0.9, 12, 0.95)
    assert counts[dist == 0].sum() == 0
    sigma = np.sqrt(rows * dist * (1 - dist))
    assert (np.abs(counts - rows * dist) <= 5 * sigma + 1).all()


def test_deterministic_and_graph_replay():
    """Integer accumulation: two launches agree bit for bit even when every atomic lands in a few
    histogram bins (flat logits); a captured launch replays with new uniforms."""
    from chitu_amd import sampling

    rows, vocab = 16, 129280
    x = _rows("flat", rows, vocab, 9).cuda()
    temps = torch.full((rows,), 0.8, device="cuda")
    ks = torch.full((rows,), -1, dtype=torch.int32, device="cuda")
    ps = torch.full((rows,), 0.9, device="cuda")
    us = torch.rand(rows, device="cuda")
    a = sampling.top_k_top_p_sampling_from_logits(x, temps, ks, ps, uniforms=us, return_stats=True)
    b = sampling.top_k_top_p_sampling_from_logits(x, temps, ks, ps, uniforms=us, return_stats=True)
    for p, q in zip(a, b):
        assert torch.equal(p, q)
    out = torch.zeros(rows, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        sampling.top_k_top_p_sampling_from_logits(x, temps, ks, ps, uniforms=us, out=out)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, a[0])
    us.copy_(torch.rand(rows, device="cuda"))
    want = sampling.top_k_top_p_sampling_from_logits(x, temps, ks, ps, uniforms=us)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)


def test_full_vocab_batch_timing(capsys, sampler_path):
    """[16, 129280] fp32 logits (one R1 decode step's sampler input): sanity + microseconds per launch."""
    from chitu_amd import sampling

    rows, vocab = 16, 129280
    x = _rows("peaked", rows, vocab, 2).cuda()
    temps = torch.full((rows,), 0.8, device="cuda")
    ks = torch.full((rows,), 50, dtype=torch.int32, device="cuda")
    ps = torch.full((rows,), 0.9, device="cuda")
    us = torch.rand(rows, device="cuda")
    out = torch.zeros(rows, dtype=torch.int64, device="cuda")
    res = {}
    for name, fn in (("sample", lambda: sampling.top_k_top_p_sampling_from_logits(x, temps, ks, ps, uniforms=us, out=out)),
                     ("argmax", lambda: sampling.argmax(x, out=out))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()  # 20 launches in one graph: device time, not the Python call overhead
        with torch.cuda.graph(graph):
            for _ in range(20):
                fn()
        graph.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20 * 1e3
    assert int(out.min()) >= 0 and int(out.max()) < vocab
    assert np.array_equal(out.cpu().numpy(), np.argmax(x.cpu().numpy(), axis=-1))
    with capsys.disabled():
        print(f"\n[sampler/{sampler_path}] rows=16 vocab=129280: sample {res['sample']:.1f} us, argmax {res['argmax']:.1f} us per launch")


# ---------------------------------------------------------------- DeviceSampler (generate()'s token selection)
@pytest.mark.parametrize("greedy", [True, False])
def test_device_sampler_bookkeeping_matches_the_reference_recipe(greedy):
    """Five steps of random logits through DeviceSampler vs executor.py:82-112 step by step: the oracle's
    frequency penalty over the tokens generated so far (bit-exact logits), then argmax / the sampling
    launch on those logits with the same uniform stream."""
    from chitu_amd import sampling

    rows, vocab, steps = 6, 777, 5
    pens = [0.5, 0.0, 2.0, 1.25, 0.0, 0.75]
    ks = [1] * rows if greedy else [1, 5, 40, -1, 3, 50]
    temps, ps = [0.7, 1.0, 1.3, 0.9, 1.0, 0.6], [1.0, 0.9, 0.8, 0.95, 0.5, 1.0]
    gen = torch.Generator(device="cuda").manual_seed(11)
    gen_ref = torch.Generator(device="cuda").manual_seed(11)
    pick = sampling.DeviceSampler(rows, steps, "cuda", temps, ks, ps, pens, gen)
    assert pick.greedy == greedy
    responses = [[] for _ in range(rows)]
    g = torch.Generator().manual_seed(3)
    for step in range(steps):
        base = torch.round(torch.randn(rows, vocab, generator=g) * 4) / 4  # coarse values: repeated tokens happen
        dev = base.clone().cuda()
        tok = pick(dev)
        want_logits = osmp.frequency_penalty(base.clone(), responses, pens, is_decode=step > 0)
        assert np.array_equal(dev.cpu().numpy().view(np.uint32), want_logits.numpy().view(np.uint32)), step
        if greedy:
            want = sampling.argmax(want_logits.cuda())
            assert np.array_equal(want.cpu().numpy(), np.argmax(want_logits.numpy(), axis=-1))
        else:
            u = torch.rand(rows, dtype=torch.float32, device="cuda", generator=gen_ref)
            want = sampling.top_k_top_p_sampling_from_logits(want_logits.cuda(), temps, ks, ps, uniforms=u)
        assert torch.equal(tok, want), step
        for r in range(rows):
            responses[r].append(int(tok[r]))
    assert pick.n_generated == steps


def test_generate_with_sampling_parameters():
    """generate(): all top_k == 1 is the greedy run; a sampling run is reproducible from the generator's
    seed, respects top_k = 1 rows, and the frequency penalty changes a greedy run that repeats itself."""
    from tests.test_gpu_deepseek import build, tiny_args

    model, _ = build(tiny_args())
    prompts = [[5, 6, 7, 8], [100], [9, 10]]
    greedy = model.generate(prompts, 6)
    assert torch.equal(greedy, model.generate(prompts, 6, top_ks=[1, 1, 1], temperatures=[0.5, 1.0, 2.0], top_ps=[0.9] * 3))

    def run(seed):
        gen = torch.Generator(device="cuda").manual_seed(seed)
        return model.generate(prompts, 6, temperatures=[1.0, 1.5, 1.0], top_ks=[50, -1, 1], top_ps=[0.95, 0.9, 1.0],
                              generator=gen)

    a, b, c = run(1), run(1), run(2)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert torch.equal(a[2], greedy[2])  # the top_k == 1 request stays greedy inside a sampling batch
    assert int(a.min()) >= 0 and int(a.max()) < model.args.vocab_size
    pen = model.generate(prompts, 6, frequency_penalties=[100.0, 100.0, 0.0])
    assert torch.equal(pen[2], greedy[2]) and torch.equal(pen[:, 0], greedy[:, 0])
    for r in (0, 1):  # a huge penalty forbids repeating any generated token
        assert len(set(pen[r].tolist())) == 6


def test_read_only_launches_accept_a_strided_vocabulary_axis():
    """The library all-gather hands back a permuted view (tensor_parallel.all_gather_last_dim without a cast): greedy and
    sampling launches take a dense copy of it, the in-place penalty refuses it."""
    from chitu_amd import sampling

    g = torch.Generator().manual_seed(3)
    dense = torch.randn(5, 1000, generator=g).cuda()
    strided = dense.t().contiguous().t()  # same values, vocabulary stride 5
    assert strided.stride(-1) != 1
    assert torch.equal(sampling.argmax(strided), sampling.argmax(dense))
    u = torch.rand(5, generator=g).cuda()
    a = sampling.top_k_top_p_sampling_from_logits(strided, [0.8] * 5, [50] * 5, [0.9] * 5, uniforms=u)
    b = sampling.top_k_top_p_sampling_from_logits(dense, [0.8] * 5, [50] * 5, [0.9] * 5, uniforms=u)
    assert torch.equal(a, b)
    with pytest.raises(Exception):
        sampling.apply_frequency_penalty(strided, [[1]] * 5, [0.5] * 5)
