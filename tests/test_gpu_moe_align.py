"""HIP moe_align vs the oracle / the reference-generated fixtures: bit-exact (integers)."""

import numpy as np
import pytest
import torch

from oracle import moe_align as oalign
from tests.util import golden

pytestmark = pytest.mark.gpu


def run_hip(ids, block, E, via="public"):
    from chitu_amd import fused_moe

    ids_d = ids.cuda()
    if via == "public":
        s, e, n = fused_moe.moe_align_block_size(ids_d, block, E)
        return s.cpu().numpy(), e.cpu().numpy(), n.cpu().numpy(), None
    # the 7-argument chitu_backend entry point, buffers pre-filled like fused_moe.py:489-506
    numel = ids.numel()
    cap = numel + E * (block - 1)
    s = torch.full((cap,), numel, dtype=torch.int32, device="cuda")
    e = torch.zeros(((cap + block - 1) // block,), dtype=torch.int32, device="cuda")
    n = torch.empty((1,), dtype=torch.int32, device="cuda")
    c = torch.zeros((E + 1,), dtype=torch.int32, device="cuda")
    fused_moe.cuda_moe_align_block_size(ids_d, E, block, s, e, n, c)
    return s.cpu().numpy(), e.cpu().numpy(), n.cpu().numpy(), c.cpu().numpy()


@pytest.mark.parametrize("case", ["doc", "reftest", "r1_bs16", "skew_b16"])
@pytest.mark.parametrize("via", ["public", "backend"])
def test_golden_fixtures(case, via):
    g = golden("moe_align")
    block, E = g[f"{case}_cfg"].tolist()
    s, e, n, _ = run_hip(torch.from_numpy(g[f"{case}_ids"]), block, E, via)
    assert np.array_equal(s, g[f"{case}_sorted"])
    assert np.array_equal(e, g[f"{case}_experts"])
    assert np.array_equal(n, g[f"{case}_npost"])


@pytest.mark.parametrize(
    "numel,E,block,dtype",
    [
        (0, 8, 4, torch.int64),       # empty
        (1, 1, 1, torch.int64),       # degenerate
        (8, 256, 64, torch.int64),    # R1 bs=1
        (8 * 32, 256, 16, torch.int64),
        (1000, 256, 64, torch.int32),  # reference test size
        (1024, 64, 128, torch.int64),  # exactly one scatter round
        (1025, 64, 128, torch.int64),  # two rounds
        (5000, 7, 3, torch.int16),     # ragged, many rounds, odd block
        (4096, 1024, 8, torch.int64),  # max experts (needs >48 KiB LDS)
        (300, 200, 32, torch.uint8),
        (18432, 257, 64, torch.int64),  # a 2048-token R1 prompt: 8 routed + 1 shared slot per token
        (63, 257, 64, torch.int64),     # fewer ids than one wave's run
        (1024 * 64 + 1, 16, 64, torch.int32),  # runs of 4160 ids per wave, the last one ragged
    ],
)
def test_against_oracle(numel, E, block, dtype):
    g = torch.Generator().manual_seed(numel * 7 + E)
    ids = torch.randint(0, E, (numel,), generator=g).to(dtype)
    s_ref, e_ref, n_ref, c_ref = oalign.moe_align_block_size(ids.numpy(), block, E)
    for via in ("public", "backend"):
        s, e, n, c = run_hip(ids, block, E, via)
        assert np.array_equal(s, s_ref), via
        assert np.array_equal(e, e_ref), via
        assert np.array_equal(n, n_ref), via
        if c is not None:
            assert np.array_equal(c, c_ref)


def test_all_tokens_one_expert_and_reference_membership_property():
    # skew: every token routed to expert 3; then the reference test's own property
    # (test/pytest/test_moe_align.py:52-74): each index appears inside its expert's segment.
    ids = torch.full((777,), 3, dtype=torch.int64)
    s, e, n, c = run_hip(ids, 64, 16, "backend")
    assert n[0] == 832 and (s[:777] == np.arange(777)).all() and (s[777:832] == 777).all()
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, 256, (1000,), generator=g)
    s, e, n, c = run_hip(ids, 64, 256, "backend")
    for ex in ids.unique().tolist():
        seg = set(s[c[ex] : c[ex + 1]].tolist())
        assert set(torch.nonzero(ids == ex).flatten().tolist()) <= seg


def test_expert_map_and_determinism():
    from chitu_amd import fused_moe

    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 32, (16, 4), generator=g).cuda()
    emap = torch.full((32,), -1, dtype=torch.int32, device="cuda")
    emap[8:16] = torch.arange(8, dtype=torch.int32, device="cuda")
    s, e, n = fused_moe.moe_align_block_size(ids, 16, 32, emap)
    s0, e0, n0 = fused_moe.moe_align_block_size(ids, 16, 32)
    assert torch.equal(e, emap[e0]) and torch.equal(s, s0)
    for _ in range(20):  # stable => run-to-run identical (the CUDA kernel is not)
        s1, e1, n1 = fused_moe.moe_align_block_size(ids, 16, 32)
        assert torch.equal(s1, s0) and torch.equal(e1, e0) and torch.equal(n1, n0)


@pytest.mark.parametrize("cfg", [
    dict(E=256, groups=(8, 4), topk=8, extra=1, score="sigmoid", bias=True),     # R1: fast routing kernel
    dict(E=64, groups=(1, 1), topk=6, extra=2, score="softmax", bias=False),     # V2-Lite: generic kernel
    dict(E=16, groups=(4, 2), topk=4, extra=1, score="sigmoid", bias=True),      # test-size V3
    dict(E=8, groups=(1, 1), topk=2, extra=0, score="softmax_renorm", bias=False),  # Mixtral-style, no extra slot
    dict(E=256, groups=(8, 4), topk=8, extra=0, score="sigmoid", bias=True, ep=(2, 8)),  # expert-parallel rank 2 of 8
], ids=["r1", "v2lite", "tiny_v3", "mixtral", "r1_ep"])
@pytest.mark.parametrize("M", [1, 2, 16, 19, 64])
def test_route_and_align_in_one_launch_equals_the_two_launches(cfg, M):
    """chitu_hip_gate_route_align (last routing workgroup sorts) == gate_route + moe_align_block_size, bit for
    bit, across repeated launches (the ticket resets itself) and under hipGraph replay with new inputs."""
    from chitu_amd import fused_moe, ops

    E, K = cfg["E"], 1024  # 16 K-splits: the routing kernel sums fp32 partials
    g = torch.Generator().manual_seed(E + M)
    w = (torch.randn(E, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    bias = (torch.randn(E, generator=g) * 0.01).to(torch.bfloat16).cuda() if cfg["bias"] else None
    n_align = E + cfg["extra"]
    emap = None
    if "ep" in cfg:
        r, ep = cfg["ep"]
        emap = torch.full((E,), -1, dtype=torch.int32)
        emap[r * (E // ep):(r + 1) * (E // ep)] = torch.arange(E // ep, dtype=torch.int32)
        emap = emap.cuda()
    kw = dict(extra_expert_id=E if cfg["extra"] else -1, extra_count=max(cfg["extra"], 1))

    def run(x, fused):
        if fused:
            return ops.gate_deepseek_v3(x, w, bias, *cfg["groups"], cfg["topk"], cfg["score"], 2.5,
                                        align=(n_align, 16, emap), **kw)
        wt, ids = ops.gate_deepseek_v3(x, w, bias, *cfg["groups"], cfg["topk"], cfg["score"], 2.5, **kw)
        return wt, ids, fused_moe.moe_align_block_size(ids, 16, n_align, emap)

    def same(a, b):
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        for p, q in zip(a[2], b[2]):
            assert p.shape == q.shape and torch.equal(p, q)

    xs = [torch.randn(M, K, generator=g).to(torch.bfloat16).cuda() for _ in range(6)]
    for x in xs:
        same(run(x, True), run(x, False))
    x_static = xs[0].clone()
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = run(x_static, True)
    for x in xs[1:4]:
        x_static.copy_(x)
        graph.replay()
        torch.cuda.synchronize()
        same(out, run(x, False))


@pytest.mark.parametrize("M", [1, 5, 16])
def test_in_routing_sort_equals_general_sort(M):
    """The one-workgroup route + align launch with the sort's order-free half inside the routing (token masks by LDS
    atomic OR, one scan, ranks by popcount: moe_align_small_*) against the same launch with the general sort behind the
    routing barrier (debug option gate_small_sort = 0): every output identical, R1 router with the shared slot and the
    expert-parallel map."""
    from chitu_amd import _lib, ops

    E, K = 256, 1024
    g = torch.Generator().manual_seed(77 + M)
    w = (torch.randn(E, K, generator=g) * K ** -0.5).to(torch.bfloat16).cuda()
    bias = (torch.randn(E, generator=g) * 0.01).to(torch.bfloat16).cuda()
    emap = torch.full((E,), -1, dtype=torch.int32)
    emap[64:96] = torch.arange(32, dtype=torch.int32)
    for extra, em in ((1, None), (0, emap.cuda())):
        kw = dict(extra_expert_id=E if extra else -1, extra_count=max(extra, 1))
        for _ in range(3):
            x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
            a = ops.gate_deepseek_v3(x, w, bias, 8, 4, 8, "sigmoid", 2.5, align=(E + extra, 16, em), **kw)
            with _lib.debug_option("gate_small_sort", 0):
                b = ops.gate_deepseek_v3(x, w, bias, 8, 4, 8, "sigmoid", 2.5, align=(E + extra, 16, em), **kw)
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
            for p, q in zip(a[2], b[2]):
                assert torch.equal(p, q)
