"""Oracle for the token sampler (frequency penalty, greedy, top-k / top-p sampling).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Restates the reference's
  chitu/executor.py:82-112  NormalExecutor.update_response (penalty, argmax / softmax + sampling)
  chitu/utils.py:62-81      top_k_top_p_min_p_sampling_from_probs_torch (sort, cumsum, two masks)
Pinned by tests/golden/sampler.npz: the reference's own update_response run here on CPU with
torch.multinomial intercepted (tests/golden/gen_sampler.py) -- the penalised logits, the masked
sorted probabilities handed to multinomial and the greedy tokens are reproduced bit-exactly by
`frequency_penalty`, `masked_sorted_probs` and `greedy`.

`sample_fixed_point` is the integer specification of chitu_hip_sample (csrc/sample.hip): same kept
prefix rule (position < top_k and exclusive cumsum <= top_p * Z) evaluated on weights scaled by
2^40 and truncated, ties in the descending order broken by lower index, the draw an inverse CDF in
index order for a given uniform u.  In `probs_mode` every operation of the kernel is an IEEE
operation, so the kernel must match it bit for bit; in logits mode the kernel's exp differs from
numpy's by an ulp and the tests allow the boundary to move by elements of negligible mass.
"""

import numpy as np
import torch

FIX_SHIFT = 40


def frequency_penalty(logits, responses, penalties, is_decode=True):
    """executor.py:89-102: logits[row].index_add_(-1, response_tokens, -penalty * ones) for rows with
    penalty > 0, a decode task and a non-empty response.  logits: float tensor [rows, vocab]
    (modified in place and returned); responses: list of int lists."""
    for it, (resp, pen) in enumerate(zip(responses, penalties)):
        if pen > 0 and is_decode and len(resp) > 0:
            idx = torch.tensor(list(resp), dtype=torch.int64)
            logits[it].index_add_(-1, idx, -pen * torch.ones((len(resp),), dtype=logits.dtype))
    return logits


def greedy(logits):
    """executor.py:103-104."""
    return torch.argmax(logits, dim=-1)


def masked_sorted_probs(probs, top_ks, top_ps):
    """utils.py:69-78 up to (not including) the multinomial draw: returns (probs_sort after both
    masks and the division by the row maximum, probs_idx)."""
    probs_sort, probs_idx = probs.sort(dim=-1, descending=True)
    probs_sum = torch.cumsum(probs_sort, dim=-1)
    probs_sort[(probs_sum - probs_sort) > top_ps.view(-1, 1)] = 0.0
    probs_sort[torch.arange(0, probs.shape[-1]).view(1, -1) >= top_ks.view(-1, 1)] = 0.0
    probs_sort.div_(probs_sort.max(dim=-1, keepdim=True)[0])
    return probs_sort, probs_idx


def softmax_probs(logits, temperatures):
    """executor.py:106."""
    return torch.softmax(logits / temperatures.view(-1, 1), dim=-1)


def _weights_fixed(row, temperature, probs_mode):
    """float32 weights e in [0, 1] relative to the row maximum and their 2^40 fixed-point form,
    with the kernel's operation order (sample_x / sample_e / sample_fix in csrc/sample.hip)."""
    v = np.asarray(row, dtype=np.float32)
    if probs_mode:
        x = v
        m = np.float32(x.max())
        with np.errstate(divide="ignore", invalid="ignore"):
            e = (x / m).astype(np.float32)
    else:
        x = (v / np.float32(temperature)).astype(np.float32)
        m = np.float32(x.max())
        e = np.exp((x - m).astype(np.float32)).astype(np.float32)
    e = np.where(np.isnan(e), np.float32(0), e)
    e = np.minimum(np.maximum(e, np.float32(0)), np.float32(1)).astype(np.float32)
    fix = np.floor(e.astype(np.float64) * float(1 << FIX_SHIFT)).astype(np.uint64)  # exact: power-of-two scale
    return x, e, fix


def kept_set_fixed_point(row, temperature, top_k, top_p, probs_mode=False):
    """(kept mask in index order, n_kept, kept integer weights, Z) under the kernel's rule."""
    x, e, fix = _weights_fixed(row, temperature, probs_mode)
    vocab = e.size
    keys = e.view(np.uint32).astype(np.int64)
    order = np.argsort(-keys, kind="stable")  # descending weight, ties by lower index
    fs = [int(f) for f in fix[order]]
    z = sum(fs)
    if top_p >= 1.0:
        p = None
    else:
        p = int(np.floor(np.float64(max(np.float32(top_p), np.float32(0))) * np.float64(z)))
    k_lim = vocab if top_k <= 0 else int(top_k)
    n = 0
    cum = 0
    for pos in range(vocab):
        if pos >= k_lim or (p is not None and cum > p):
            break
        n += 1
        cum += fs[pos]
    mask = np.zeros(vocab, dtype=bool)
    mask[order[:n]] = True
    return mask, n, fix, z


def sample_fixed_point(row, temperature, top_k, top_p, u, probs_mode=False):
    """Token chitu_hip_sample must return for this row, its n_kept and kept mass fraction."""
    x, e, fix = _weights_fixed(row, temperature, probs_mode)
    if top_k == 1:
        return _argmax(x), 1, 0.0
    mask, n, fix, z = kept_set_fixed_point(row, temperature, top_k, top_p, probs_mode)
    w = [int(f) if k else 0 for f, k in zip(fix, mask)]
    s_kept = sum(w)
    if s_kept == 0:
        return _argmax(x), n, 0.0
    uu = np.float32(min(max(np.float32(u), np.float32(0)), np.float32(1)))
    t = np.float64(uu) * np.float64(s_kept)
    target = s_kept - 1 if t >= np.float64(s_kept) else int(t)
    target = min(target, s_kept - 1)
    acc = 0
    for i, wi in enumerate(w):
        acc += wi
        if acc > target:
            return i, n, s_kept / z
    raise AssertionError("unreachable")


def _argmax(x):
    """first index of the maximum, NaN counted as the largest value (torch.argmax)."""
    x = np.asarray(x, dtype=np.float32)
    nan = np.isnan(x)
    return int(np.argmax(nan)) if nan.any() else int(np.argmax(x))


def kept_distribution(row, temperature, top_k, top_p, probs_mode=False):
    """Normalised sampling distribution over the vocabulary (float64) under the kernel's rule."""
    mask, n, fix, z = kept_set_fixed_point(row, temperature, top_k, top_p, probs_mode)
    w = np.where(mask, fix.astype(np.float64), 0.0)
    return w / w.sum()


def sample_radix_model(row, temperature, top_k, top_p, u, probs_mode=False, waves=16):
    """The same answer computed the way csrc/sample.hip computes it -- three 10-bit histogram levels
    over the weight's float pattern, boundary bin per level, tie quota at tau, per-wave segment
    sums, inverse CDF inside one segment -- so the kernel's control flow is checked on the CPU
    against `sample_fixed_point` (tests/test_oracle_golden.py)."""
    x, e, fix = _weights_fixed(row, temperature, probs_mode)
    vocab = e.size
    arg_max = _argmax(x)
    if top_k == 1:
        return arg_max, 1
    keys = e.view(np.uint32).astype(np.int64)
    fixl = [int(f) for f in fix]
    k_rem = (1 << 32) - 1 if top_k <= 0 else int(top_k)
    p_rem = None
    tau, c_above, s_above, e_tau, c_keep, keep_all = 0, 0, 0, 0, (1 << 32) - 1, False
    prefix, z_total = 0, 0
    for level, shift in enumerate((20, 10, 0)):
        cnt = [0] * 1024
        sm = [0] * 1024
        for i in range(vocab):
            k = int(keys[i])
            if level > 0 and (k >> (shift + 10)) != prefix:
                continue
            b = (k >> shift) & 1023
            cnt[b] += 1
            sm[b] += fixl[i]
        tot = sum(sm)
        if level == 0:
            z_total = tot
            if top_p >= 1.0:
                p_rem = (1 << 64) - 1
            else:
                p_rem = int(np.floor(np.float64(max(np.float32(top_p), np.float32(0))) * np.float64(tot)))
        found = None
        c_incl, s_incl = 0, 0
        for b in range(1023, -1, -1):
            c_excl, s_excl = c_incl, s_incl
            c_incl += cnt[b]
            s_incl += sm[b]
            cond = c_incl >= k_rem or s_incl > p_rem
            cond_prev = c_excl >= k_rem or s_excl > p_rem
            if cond and not cond_prev:
                assert found is None
                found = (b, c_excl, s_excl, cnt[b], sm[b])
        if found is None:
            assert level == 0, "a child of a boundary bin must hold the boundary"
            keep_all = True
            break
        b, c_excl, s_excl, c_b, s_b = found
        prefix = (prefix << 10) | b if level > 0 else b
        c_above += c_excl
        s_above += s_excl
        k_rem -= c_excl
        p_rem -= s_excl
        if level == 2:
            tau = prefix
            e_tau = s_b // c_b
            assert e_tau * c_b == s_b
            c_p = min(p_rem // e_tau + 1, c_b) if e_tau else c_b
            c_keep = min(c_p, k_rem)
    s_kept = z_total if keep_all else s_above + c_keep * e_tau
    n_kept = vocab if keep_all else c_above + c_keep
    if s_kept == 0:
        return arg_max, n_kept
    uu = np.float32(min(max(np.float32(u), np.float32(0)), np.float32(1)))
    t = np.float64(uu) * np.float64(s_kept)
    target = s_kept - 1 if t >= np.float64(s_kept) else min(int(t), s_kept - 1)
    seg_len = ((vocab + waves - 1) // waves + 255) // 256 * 256
    cum, tb, sel = 0, 0, None
    for w in range(waves):
        lo, hi = min(w * seg_len, vocab), min(w * seg_len + seg_len, vocab)
        strict = sum(fixl[i] for i in range(lo, hi) if int(keys[i]) > tau)
        ties = sum(1 for i in range(lo, hi) if int(keys[i]) == tau)
        kt = min(max(c_keep - tb, 0), ties)
        s = strict + kt * e_tau
        if sel is None and target < cum + s:
            sel = (lo, hi, cum, tb)
        cum += s
        tb += ties
    assert cum == s_kept and sel is not None
    lo, hi, run, rank = sel
    for i in range(lo, hi):
        k = int(keys[i])
        w = fixl[i] if k > tau else 0
        if k == tau:
            if rank < c_keep:
                w = e_tau
            rank += 1
        run += w
        if run > target:
            return i, n_kept
    raise AssertionError("unreachable")


def sample_candidate_model(row, temperature, top_k, top_p, u, probs_mode=False, cap=1024):
    """The candidate fast path of csrc/sample.hip (sample_candidates) restated: threshold counts at
    2^-4 .. 2^-20 of the maximum, the lowest threshold admitting <= cap candidates, the cut on the sorted
    candidates, the sufficiency rule, the inverse CDF over the kept entries in index order.  Returns
    (token, n_kept), or None where the kernel falls back to the radix descent -- so the rule that decides
    when the short list provably holds the whole kept prefix is checked on the CPU against
    `sample_fixed_point` (tests/test_sampler_oracle.py) with caps small enough to hit every branch."""
    x, e, fix = _weights_fixed(row, temperature, probs_mode)
    vocab = e.size
    if top_k == 1:
        return _argmax(x), 1
    keys = e.view(np.uint32).astype(np.int64)
    fixl = [int(f) for f in fix]
    z = sum(fixl)
    tkeys = [(127 - 4 * (j + 1)) << 23 for j in range(5)]
    counts = [int((keys >= t).sum()) for t in tkeys]
    if z == 0 or counts[0] > cap:
        return None
    level = max(j for j in range(5) if counts[j] <= cap)
    cand = [i for i in range(vocab) if int(keys[i]) >= tkeys[level]]
    cand.sort(key=lambda i: (-int(keys[i]), i))
    if top_p >= 1.0:
        p_rem = (1 << 64) - 1
    else:
        p_rem = int(np.floor(np.float64(max(np.float32(top_p), np.float32(0))) * np.float64(z)))
    k_eff = (1 << 32) - 1 if top_k <= 0 else int(top_k)
    kept, run = [], 0
    for pos, i in enumerate(cand):
        if pos < k_eff and run <= p_rem:
            kept.append(i)
        run += fixl[i]
    s_cand = run
    n_cand, n_kept = len(cand), len(kept)
    if not (n_kept < n_cand or n_cand >= k_eff or s_cand > p_rem or n_cand == vocab):
        return None
    s_kept = sum(fixl[i] for i in kept)
    if s_kept == 0:
        return _argmax(x), n_kept
    uu = np.float32(min(max(np.float32(u), np.float32(0)), np.float32(1)))
    t = np.float64(uu) * np.float64(s_kept)
    target = s_kept - 1 if t >= np.float64(s_kept) else min(int(t), s_kept - 1)
    run = 0
    for i in sorted(kept):
        run += fixl[i]
        if run > target:
            return i, n_kept
    raise AssertionError("unreachable")
