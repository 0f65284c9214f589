"""INT8 W8A8 fused MoE (BASELINE config 4 building block) vs the per-expert W8A8Linear loop oracle."""

import pytest
import torch

from oracle import w8a8 as ow
from tests.util import assert_close, max_rel_to_peak

pytestmark = pytest.mark.gpu


def make_case(M, E, topk, K, I, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, K, generator=g) * 0.7).to(torch.bfloat16)
    w1f = torch.randn(E, 2 * I, K, generator=g) * K ** -0.5
    w2f = torch.randn(E, K, I, generator=g) * I ** -0.5
    w1, s1 = zip(*(ow.quant_weight(w) for w in w1f))
    w2, s2 = zip(*(ow.quant_weight(w) for w in w2f))
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)])
    wts = torch.softmax(torch.randn(M, topk, generator=g), -1).to(torch.bfloat16)
    return x, torch.stack(w1), torch.stack(w2), torch.stack(s1), torch.stack(s2), ids, wts


def run_hip(x, w1, w2, s1, s2, ids, wts, **kw):
    from chitu_amd import fused_moe

    return fused_moe.fused_experts(x.cuda().clone(), w1.cuda(), w2.cuda(), wts.cuda(), ids.cuda(), use_int8_w8a8=True,
                                   w1_scale=s1.cuda(), w2_scale=s2.cuda(), **kw).cpu()


@pytest.mark.parametrize("M,E,topk,K,I", [(1, 8, 2, 4096, 3584), (16, 8, 2, 4096, 3584), (33, 8, 2, 512, 256), (5, 16, 4, 256, 128),
                                          (70, 4, 2, 1024, 384)])
def test_vs_per_expert_w8a8_loop(M, E, topk, K, I):
    args = make_case(M, E, topk, K, I, seed=M + E)
    out = run_hip(*args)
    x, w1, w2, s1, s2, ids, wts = args
    ref = ow.fused_experts_int8(x, w1, w2, wts, ids, s1, s2)
    assert_close(out, ref, 1e-2)
    assert ((out.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item() < 5e-3


def test_determinism_expert_map_and_unreduced_view():
    args = make_case(9, 8, 2, 512, 256, seed=4)
    a = run_hip(*args)
    for _ in range(3):
        assert torch.equal(run_hip(*args), a)  # integer accumulation, no atomics
    x, w1, w2, s1, s2, ids, wts = args
    emap = torch.tensor([0, 1, 2, 3, -1, -1, -1, -1], dtype=torch.int32)
    out = run_hip(x, w1[:4].contiguous(), w2[:4].contiguous(), s1[:4].contiguous(), s2[:4].contiguous(), ids, wts,
                  expert_map=emap.cuda(), global_num_experts=8)
    masked = torch.where(ids < 4, wts.float(), torch.zeros(())).to(wts.dtype)
    assert_close(out, ow.fused_experts_int8(x, w1, w2, masked, ids, s1, s2), 1e-2)
    c3 = run_hip(*args, reduce_topk=False)
    assert tuple(c3.shape) == (9, 2, 512)
    assert torch.equal(c3.float().sum(1).to(torch.bfloat16), a)
