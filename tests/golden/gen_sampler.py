"""Golden fixture for the sampler: the REFERENCE's own NormalExecutor.update_response
(chitu/executor.py:82-112) and top_k_top_p_min_p_sampling_from_probs_torch (chitu/utils.py:62-81)
run on CPU with duck-typed tasks; torch.multinomial is intercepted to record the masked, sorted
probabilities it is handed (the deterministic part of the sampler) and returns column 0.

Run in the build container only:   python tests/golden/gen_sampler.py   -> tests/golden/sampler.npz
"""

import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()

import numpy as np  # noqa: E402
import torch  # noqa: E402

import chitu.executor as ex  # noqa: E402
import chitu.utils as ref_utils  # noqa: E402
from chitu.task import TaskType  # noqa: E402


class _Resp:
    def __init__(self, toks):
        self.toks = list(toks)

    def __len__(self):
        return len(self.toks)

    def to_tensor(self):
        return torch.tensor(self.toks, dtype=torch.int64)


class _Task:
    def __init__(self, resp, pen, top_k):
        self.response = _Resp(resp)
        self.task_type = TaskType.Decode
        self.req = types.SimpleNamespace(params=types.SimpleNamespace(frequency_penalty=pen, top_k=top_k))
        self.picked = None

    def update_response(self, tok_cpu, tok):
        self.picked = tok_cpu


def run_reference(logits, responses, penalties, temperatures, top_ks, top_ps):
    tasks = [_Task(r, p, k) for r, p, k in zip(responses, penalties, top_ks)]
    packed = types.SimpleNamespace(
        tasks=tasks, is_all_greedy=all(k <= 1 for k in top_ks), temperatures=torch.tensor(temperatures),
        top_ps=torch.tensor(top_ps), top_ks=torch.tensor(top_ks))
    captured = {}
    real_multinomial = torch.multinomial

    def fake_multinomial(p, num_samples=1, **kw):
        captured["masked"] = p.clone()
        return torch.zeros(p.shape[0], 1, dtype=torch.int64)

    torch.multinomial = fake_multinomial
    try:
        lg = logits.clone()
        ex.NormalExecutor.update_response(None, packed, lg)
    finally:
        torch.multinomial = real_multinomial
    return lg, captured.get("masked"), [t.picked for t in tasks]


def main():
    g = torch.Generator().manual_seed(0)
    vocab = 1024
    rows = 8
    logits = torch.randn(rows, vocab, generator=g) * 2.5
    logits[3] = torch.round(logits[3])  # ties
    logits[5] = logits[5] * 0.01  # flat
    responses = [[], [5, 5, 9, 1023], [7], [1, 2, 3, 3, 3], [], [0, 0], [100, 200, 100], [11]]
    penalties = [0.1, 0.5, 0.0, 1.25, 0.3, -0.2, 2.0, 0.1]
    temperatures = [0.8, 1.0, 0.5, 1.3, 0.7, 1.0, 2.0, 0.9]
    top_ks = [50, 5, 1000, 20, 1, 300, 2000, 3]
    top_ps = [0.9, 0.95, 0.5, 1.0, 0.9, 0.8, 0.3, 0.99]
    pen_logits, masked, picked = run_reference(logits, responses, penalties, temperatures, top_ks, top_ps)
    # the all-greedy branch (every top_k <= 1)
    g_logits, g_masked, g_picked = run_reference(logits, responses, penalties, temperatures, [1] * rows, top_ps)
    assert g_masked is None
    # the stand-alone probability-space function on the penalised logits' softmax
    probs = torch.softmax(pen_logits / torch.tensor(temperatures).view(-1, 1), dim=-1)
    out = dict(
        logits=logits.numpy(), resp_flat=np.array([t for r in responses for t in r], dtype=np.int64),
        resp_off=np.cumsum([0] + [len(r) for r in responses]).astype(np.int64),
        penalties=np.array(penalties, dtype=np.float32), temperatures=np.array(temperatures, dtype=np.float32),
        top_ks=np.array(top_ks, dtype=np.int64), top_ps=np.array(top_ps, dtype=np.float32),
        penalised_logits=pen_logits.numpy(), masked_sorted=masked.numpy(),
        picked_col0=np.array(picked, dtype=np.int64), greedy_tokens=np.array(g_picked, dtype=np.int64),
        probs=probs.numpy())
    path = os.path.join(HERE, "sampler.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})
    assert ref_utils.top_k_top_p_min_p_sampling_from_probs_torch is not None


if __name__ == "__main__":
    main()
