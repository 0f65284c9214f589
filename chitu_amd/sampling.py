"""Token sampling on the GPU: the step right after the decode hot path.

Reference (read-only): `NormalExecutor.update_response` chitu/executor.py:82-112 (frequency penalty,
argmax for all-greedy batches, else softmax(logits / temperature) + top-k / top-p sampling) and
`top_k_top_p_min_p_sampling_from_probs_torch` chitu/utils.py:62-81 (full sort + cumsum + multinomial).
Here: `chitu_hip_frequency_penalty` + ONE `chitu_hip_sample` launch (csrc/sample.hip, no sort), tokens
stay on the device, everything is hipGraph-capturable.

Semantics kept: penalty only for rows with frequency_penalty > 0 and a non-empty response; a batch is
greedy iff every top_k <= 1; a top_k of 1 inside a sampling batch keeps only the arg-max; an entry at
sorted position `pos` survives iff pos < top_k and its exclusive cumulative probability <= top_p.
Conscious differences: a top_k <= 0 inside a sampling batch means "no top-k limit" (serve.py:52
documents -1 that way; the reference's mask zeroes the whole row and multinomial fails); ties in the
sorted order go to the lower token id (torch.sort leaves them unspecified); the draw is an inverse CDF
driven by one uniform per row from a torch.Generator (multinomial's exponential-race draw has the same
distribution but another stream).
"""

from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check, float_dtype_code, i32, i64, ptr, require_cuda, stream_ptr

__all__ = [
    "apply_frequency_penalty",
    "argmax",
    "top_k_top_p_sampling_from_logits",
    "top_k_top_p_min_p_sampling_from_probs_torch",
    "sample_tokens",
    "DeviceSampler",
]


def _rows(t: torch.Tensor, writable: bool = False) -> Tuple[torch.Tensor, int, int]:
    """[rows, vocab] with a unit stride along the vocabulary.  Read-only consumers take a dense copy of a strided input
    (e.g. the permuted view a library all-gather hands back); in-place ones refuse it."""
    assert t.dim() >= 1
    if t.stride(-1) != 1 and not writable:
        t = t.contiguous()
    t2 = t.reshape(-1, t.shape[-1]) if not writable else t.view(-1, t.shape[-1])
    assert t2.stride(-1) == 1, "the vocabulary axis must be contiguous"
    return t2, t2.shape[0], t2.shape[1]


def apply_frequency_penalty(logits: torch.Tensor, responses: Sequence[Sequence[int]],
                            frequency_penalties: Sequence[float]) -> torch.Tensor:
    """logits[row, t] -= penalty[row] for every occurrence of t in responses[row], in place
    (executor.py:89-102).  Rows with penalty <= 0 or an empty response are untouched."""
    require_cuda(logits)
    assert logits.dtype == torch.float32, "fp32 logits (model.py:475 casts them)"
    lg, rows, vocab = _rows(logits, writable=True)
    assert len(responses) == rows and len(frequency_penalties) == rows
    if not any(p > 0 and len(r) > 0 for p, r in zip(frequency_penalties, responses)):
        return logits
    flat: List[int] = []
    offs = [0]
    for r in responses:
        flat.extend(int(t) for t in r)
        offs.append(len(flat))
    dev = logits.device
    tok = torch.tensor(flat if flat else [0], dtype=torch.int32, device=dev)
    off = torch.tensor(offs, dtype=torch.int32, device=dev)
    pen = torch.tensor([float(p) for p in frequency_penalties], dtype=torch.float32, device=dev)
    return apply_frequency_penalty_device(logits, tok, off, pen)


def apply_frequency_penalty_device(logits, tokens_i32, offsets_i32, penalties_f32):
    """Same with the ragged response lists already on the device (graph-capturable): tokens_i32 flat,
    offsets_i32 [rows + 1], penalties_f32 [rows]."""
    require_cuda(logits, tokens_i32, offsets_i32, penalties_f32)
    lg, rows, vocab = _rows(logits, writable=True)
    assert logits.dtype == torch.float32
    assert tokens_i32.dtype == torch.int32 and offsets_i32.dtype == torch.int32 and penalties_f32.dtype == torch.float32
    assert offsets_i32.numel() == rows + 1 and penalties_f32.numel() == rows
    check(_lib.lib().chitu_hip_frequency_penalty(ptr(lg), i64(lg.stride(0)), i64(rows), i32(vocab), ptr(tokens_i32),
                                                 ptr(offsets_i32), ptr(penalties_f32), stream_ptr()),
          "frequency_penalty")
    return logits


def _launch(x, temperatures, top_ks, top_ps, uniforms, probs_mode, out, stats):
    x2, rows, vocab = _rows(x)
    n_kept = mass = None
    if stats:
        n_kept = torch.empty(rows, dtype=torch.int32, device=x.device)
        mass = torch.empty(rows, dtype=torch.float32, device=x.device)
    if out is None:
        out = torch.empty(rows, dtype=torch.int64, device=x.device)
    assert out.dtype == torch.int64 and out.numel() == rows and out.is_contiguous()
    check(_lib.lib().chitu_hip_sample(ptr(x2), float_dtype_code(x.dtype), i64(x2.stride(0)), i64(rows), i32(vocab),
                                      ptr(temperatures), ptr(top_ks), ptr(top_ps), ptr(uniforms), i32(probs_mode),
                                      ptr(out), ptr(n_kept), ptr(mass), stream_ptr()), "sample")
    return (out, n_kept, mass) if stats else out


def argmax(logits: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Greedy tokens [rows] int64 (executor.py:103-104): first index of the row maximum."""
    require_cuda(logits)
    return _launch(logits, None, None, None, None, 0, out, False)


def _per_row(v, rows, dtype, device):
    if isinstance(v, torch.Tensor):
        t = v.to(device=device, dtype=dtype).contiguous().view(-1)
    else:
        t = torch.tensor(list(v), dtype=dtype, device=device)
    assert t.numel() == rows
    return t


def _uniforms(rows, device, uniforms, generator):
    if uniforms is not None:
        return _per_row(uniforms, rows, torch.float32, device)
    return torch.rand(rows, dtype=torch.float32, device=device, generator=generator)


def top_k_top_p_sampling_from_logits(logits, temperatures, top_ks, top_ps, uniforms=None, generator=None,
                                     out=None, return_stats=False):
    """softmax(logits / temperature) -> top-k / top-p mask -> one draw per row, in one launch
    (executor.py:106-109).  `uniforms` [rows] in [0, 1) makes the draw reproducible; otherwise they
    come from `generator` (a CUDA torch.Generator) or the default one.  return_stats: also the
    number of kept tokens and their probability mass per row."""
    require_cuda(logits)
    _, rows, _ = _rows(logits)
    dev = logits.device
    return _launch(logits, _per_row(temperatures, rows, torch.float32, dev), _per_row(top_ks, rows, torch.int32, dev),
                   _per_row(top_ps, rows, torch.float32, dev), _uniforms(rows, dev, uniforms, generator), 0, out,
                   return_stats)


def top_k_top_p_min_p_sampling_from_probs_torch(probs, top_ks, top_ps, min_ps=None, uniforms=None, generator=None,
                                                return_stats=False):
    """Same name and arguments as chitu/utils.py:62-81 (min_ps is unused there too), on probabilities."""
    assert min_ps is None, "min_p is a TODO in the reference (utils.py:66) and is not implemented"
    require_cuda(probs)
    _, rows, _ = _rows(probs)
    dev = probs.device
    return _launch(probs, None, _per_row(top_ks, rows, torch.int32, dev), _per_row(top_ps, rows, torch.float32, dev),
                   _uniforms(rows, dev, uniforms, generator), 1, None, return_stats)


def sample_tokens(logits, temperatures, top_ks, top_ps, frequency_penalties=None, responses=None, uniforms=None,
                  generator=None):
    """update_response's device work (executor.py:82-109) for a decode batch: optional frequency
    penalty, then argmax if every top_k <= 1 (`is_all_greedy`, task.py:457) else the sampling launch.
    logits [rows, vocab] fp32 (penalised in place, like the reference); returns int64 tokens on the
    device (the reference's one `.cpu()` per step is the caller's choice)."""
    logits2, rows, _ = _rows(logits)
    if frequency_penalties is not None and responses is not None:
        apply_frequency_penalty(logits2, responses, frequency_penalties)
    ks = [int(k) for k in (top_ks.tolist() if isinstance(top_ks, torch.Tensor) else top_ks)]
    if all(k <= 1 for k in ks):
        return argmax(logits2)
    return top_k_top_p_sampling_from_logits(logits2, temperatures, ks, top_ps, uniforms=uniforms, generator=generator)


class DeviceSampler:
    """Sampling state of one generate() call kept on the device: the per-request parameters
    (task.py:433-457 gathers them from the requests every step), the tokens generated so far (the
    frequency penalty's input, executor.py:89-102) and the uniform stream.  Calling it with the step's
    fp32 logits [n_req, vocab] returns the int64 tokens [n_req] without a host round trip: at most one
    penalty launch, one sampling launch, and the bookkeeping copies.

    Greedy iff every top_k <= 1 (`is_all_greedy`, task.py:457) -- also the default when no sampling
    parameter is given.  `generator`: a CUDA torch.Generator for the uniforms (reproducible runs)."""

    def __init__(self, n_req: int, max_new_tokens: int, device, temperatures=None, top_ks=None, top_ps=None,
                 frequency_penalties=None, generator=None):
        self.n_req, self.device, self.generator = n_req, torch.device(device), generator
        ks = [1] * n_req if top_ks is None else [int(k) for k in (top_ks.tolist() if isinstance(top_ks, torch.Tensor) else top_ks)]
        assert len(ks) == n_req
        self.greedy = all(k <= 1 for k in ks)
        if not self.greedy:
            self.top_ks = _per_row(ks, n_req, torch.int32, self.device)
            self.temperatures = _per_row([1.0] * n_req if temperatures is None else temperatures, n_req, torch.float32, self.device)
            self.top_ps = _per_row([1.0] * n_req if top_ps is None else top_ps, n_req, torch.float32, self.device)
        pens = None if frequency_penalties is None else [float(p) for p in frequency_penalties]
        self.penalise = pens is not None and any(p > 0 for p in pens)
        if self.penalise:
            assert len(pens) == n_req
            self.penalties = torch.tensor(pens, dtype=torch.float32, device=self.device)
            self.history = torch.zeros(n_req, max(max_new_tokens, 1), dtype=torch.int32, device=self.device)
            self.row_ids = torch.arange(n_req + 1, dtype=torch.int32, device=self.device)
        self.n_generated = 0

    def __call__(self, logits: torch.Tensor) -> torch.Tensor:
        lg, rows, _ = _rows(logits)
        assert rows == self.n_req
        t = self.n_generated
        if self.penalise:
            assert t <= self.history.shape[1], "DeviceSampler: more calls than max_new_tokens (the penalty history is full)"
        if self.penalise and t > 0:
            # every request has generated exactly t tokens: row r's list is history[r, :t]
            flat = self.history[:, :t].contiguous().view(-1)  # (a [n, 1] slice would otherwise reshape to a strided view)
            apply_frequency_penalty_device(lg, flat, self.row_ids * t, self.penalties)
        if self.greedy:
            tok = argmax(lg)
        else:
            tok = top_k_top_p_sampling_from_logits(lg, self.temperatures, self.top_ks, self.top_ps, generator=self.generator)
        if self.penalise and t < self.history.shape[1]:
            self.history[:, t] = tok.to(torch.int32)
        self.n_generated = t + 1
        return tok
