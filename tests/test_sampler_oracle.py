"""Sampler oracle (oracle/sampling.py) pinned against the reference's own update_response run on
CPU (tests/golden/gen_sampler.py -> sampler.npz), and the kernel's integer specification checked
against both the reference's masks and a CPU model of the kernel's control flow."""

import numpy as np
import pytest
import torch

from oracle import sampling as osmp
from tests.util import golden


def _responses(g):
    off = g["resp_off"].tolist()
    flat = g["resp_flat"].tolist()
    return [flat[off[i]: off[i + 1]] for i in range(len(off) - 1)]


def test_frequency_penalty_matches_reference():
    g = golden("sampler")
    out = osmp.frequency_penalty(torch.from_numpy(g["logits"].copy()), _responses(g), g["penalties"].tolist())
    assert np.array_equal(out.numpy().view(np.uint32), g["penalised_logits"].view(np.uint32))
    # rows whose penalty is <= 0 or whose response is empty are untouched (executor.py:89-93)
    for row in (0, 2, 4, 5):
        assert np.array_equal(g["penalised_logits"][row], g["logits"][row])


def test_masks_match_reference():
    g = golden("sampler")
    probs = osmp.softmax_probs(torch.from_numpy(g["penalised_logits"]), torch.from_numpy(g["temperatures"]))
    assert np.array_equal(probs.numpy().view(np.uint32), g["probs"].view(np.uint32))
    masked, idx = osmp.masked_sorted_probs(probs, torch.from_numpy(g["top_ks"]), torch.from_numpy(g["top_ps"]))
    assert np.array_equal(masked.numpy().view(np.uint32), g["masked_sorted"].view(np.uint32))
    # multinomial was intercepted to return column 0: the reference then emits the most likely token
    assert np.array_equal(idx[:, 0].numpy(), g["picked_col0"])


def test_greedy_matches_reference():
    g = golden("sampler")
    assert np.array_equal(osmp.greedy(torch.from_numpy(g["penalised_logits"])).numpy(), g["greedy_tokens"])


def test_integer_specification_agrees_with_reference_masks():
    """The kernel's rule (integers, 2^40 scale) keeps the same tokens with the same relative
    probabilities as the reference's float masks, on the reference-generated fixture."""
    g = golden("sampler")
    probs = torch.from_numpy(g["probs"])
    _, idx = probs.sort(dim=-1, descending=True)
    masked = g["masked_sorted"]
    for row in range(probs.shape[0]):
        ref_sorted = masked[row].astype(np.float64)
        n_ref = int((ref_sorted > 0).sum())
        ref = np.zeros(probs.shape[1])
        ref[idx[row].numpy()] = ref_sorted
        ref /= ref.sum()
        for probs_mode, src, t in ((False, g["penalised_logits"][row], float(g["temperatures"][row])),
                                   (True, g["probs"][row], 1.0)):
            mask, n, fix, z = osmp.kept_set_fixed_point(src, t, int(g["top_ks"][row]), float(g["top_ps"][row]), probs_mode)
            assert n == n_ref, (row, probs_mode, n, n_ref)
            dist = osmp.kept_distribution(src, t, int(g["top_ks"][row]), float(g["top_ps"][row]), probs_mode)
            # the multiset of kept probabilities always agrees; which members of a tie group at the
            # boundary survive is unspecified in the reference (torch.sort is not stable), so the
            # per-token comparison skips the row with rounded (heavily tied) logits
            assert np.abs(np.sort(dist) - np.sort(ref)).max() < 2e-6
            if row != 3:
                assert np.abs(dist - ref).max() < 2e-6
                assert np.array_equal(dist > 0, ref > 0)


@pytest.mark.parametrize("seed", range(4))
def test_radix_descent_model_equals_specification(seed):
    """csrc/sample.hip's control flow (three histogram levels, tie quota, per-wave segments) restated
    on the CPU gives the specification's token and kept count in every regime: peaked, flat, heavy
    ties, all-equal, underflowing weights; top-k / top-p / both / neither; u at both ends."""
    rng = np.random.default_rng(seed)
    for trial in range(120):
        vocab = int(rng.choice([5, 64, 257, 1000, 4099]))
        kind = trial % 6
        logits = [rng.normal(0, 3, vocab), rng.normal(0, 0.01, vocab), np.round(rng.normal(0, 2, vocab)),
                  np.zeros(vocab), rng.normal(0, 30, vocab), rng.normal(0, 1, vocab)][kind].astype(np.float32)
        t = float(rng.choice([0.3, 0.8, 1.0, 1.7]))
        top_k = int(rng.choice([-1, 0, 2, 5, 50, vocab, vocab + 10]))
        top_p = float(rng.choice([0.0, 0.1, 0.5, 0.9, 0.999, 1.0]))
        u = [0.0, 0.99999994, float(rng.random())][min(trial % 5, 2)]
        probs_mode = trial % 4 == 0
        row = torch.softmax(torch.from_numpy(logits) / t, -1).numpy() if probs_mode else logits
        tok, n, _ = osmp.sample_fixed_point(row, t, top_k, top_p, u, probs_mode)
        tok2, n2 = osmp.sample_radix_model(row, t, top_k, top_p, u, probs_mode)
        assert (tok, n) == (tok2, n2), (trial, vocab, kind, t, top_k, top_p, u, probs_mode)


@pytest.mark.parametrize("seed", range(4))
def test_candidate_path_model_equals_specification_or_declines(seed):
    """The kernel's short-list path (thresholds, sufficiency rule, two sorts) restated on the CPU: wherever
    it answers, the answer is the specification's; caps of 4 .. 1024 make it decline, cut inside the list,
    stop at top_k, stop at top_p and take the whole row."""
    rng = np.random.default_rng(100 + seed)
    answered = declined = 0
    for trial in range(160):
        vocab = int(rng.choice([5, 64, 257, 1000, 4099]))
        kind = trial % 6
        logits = [rng.normal(0, 3, vocab), rng.normal(0, 0.01, vocab), np.round(rng.normal(0, 2, vocab)),
                  np.zeros(vocab), rng.normal(0, 30, vocab), rng.normal(0, 1, vocab)][kind].astype(np.float32)
        t = float(rng.choice([0.3, 0.8, 1.0, 1.7]))
        top_k = int(rng.choice([-1, 0, 2, 5, 50, vocab, vocab + 10]))
        top_p = float(rng.choice([0.0, 0.1, 0.5, 0.9, 0.999, 1.0]))
        u = [0.0, 0.99999994, float(rng.random())][min(trial % 5, 2)]
        probs_mode = trial % 4 == 0
        cap = int(rng.choice([4, 16, 64, 1024]))
        row = torch.softmax(torch.from_numpy(logits) / t, -1).numpy() if probs_mode else logits
        got = osmp.sample_candidate_model(row, t, top_k, top_p, u, probs_mode, cap=cap)
        if got is None:
            declined += 1
            continue
        answered += 1
        tok, n, _ = osmp.sample_fixed_point(row, t, top_k, top_p, u, probs_mode)
        assert got == (tok, n), (trial, vocab, kind, t, top_k, top_p, u, probs_mode, cap)
    assert answered > 40 and declined > 20


def test_draws_follow_the_kept_distribution():
    """Inverse CDF over the kept weights: a fine grid of u reproduces the kept distribution."""
    rng = np.random.default_rng(0)
    logits = rng.normal(0, 2, 300).astype(np.float32)
    dist = osmp.kept_distribution(logits, 0.9, 12, 0.95)
    n = 4000
    counts = np.zeros(300)
    for i in range(n):
        counts[osmp.sample_fixed_point(logits, 0.9, 12, 0.95, (i + 0.5) / n)[0]] += 1
    assert np.abs(counts / n - dist).max() < 1.0 / n + 1e-6
