"""HIP fused MoE (fp8 block-scaled W8A8) vs the oracle restatement of fused_experts_impl."""

import numpy as np
import pytest
import torch

from oracle import moe as omoe
from tests.util import assert_close, assert_close_elementwise, bf16, fp8, golden, max_rel_to_peak

pytestmark = pytest.mark.gpu

REL_TOL = 1e-2


def make_case(M, E, topk, K, I, seed, w_dtype=torch.bfloat16, dup_free=True):
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * I, K, generator=g) * 0.5).to(torch.float8_e4m3fn)
    w2 = (torch.randn(E, K, I, generator=g) * 0.5).to(torch.float8_e4m3fn)
    w1s = torch.rand(E, (2 * I + 127) // 128, K // 128, generator=g) * 0.02 + 0.01
    w2s = torch.rand(E, K // 128, I // 128, generator=g) * 0.02 + 0.01
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)])
    wts = torch.rand(M, topk, generator=g).to(w_dtype)
    return x, w1, w2, w1s, w2s, ids, wts


def run_hip(x, w1, w2, w1s, w2s, ids, wts, inplace=False, expert_map=None, global_num_experts=-1):
    from chitu_amd import fused_moe

    xd = x.cuda().clone()
    out = fused_moe.fused_experts(
        xd, w1.cuda(), w2.cuda(), wts.cuda(), ids.cuda(), inplace=inplace, use_fp8_w8a8=True,
        w1_scale=w1s.cuda(), w2_scale=w2s.cuda(), block_shape=[128, 128], expert_map=expert_map,
        global_num_experts=global_num_experts,
    )
    if inplace:
        assert out.data_ptr() == xd.data_ptr()
    return out.cpu()


def test_reference_fixture_inputs():
    g = golden("fused_moe_fp8")
    args = (bf16(g["x"]), fp8(g["w1"]), fp8(g["w2"]), torch.from_numpy(g["w1s"]), torch.from_numpy(g["w2s"]),
            torch.from_numpy(g["ids"]), bf16(g["wts"]))
    out = run_hip(*args)
    ref = omoe.fused_experts_fp8(args[0], args[1], args[2], args[6], args[5], args[3], args[4])
    assert_close(out, ref, REL_TOL)  # the bar: HIP vs the oracle with IEEE casts (what the reference computes on a GPU)
    # The fixture was produced under the Triton INTERPRETER, whose fp8 / bf16 casts are defective (tests/test_oracle_golden.py):
    # it differs from any correctly rounding implementation by ~10 % of its peak.  The loose bar below is therefore never
    # the only link to the reference's own run -- the explaining link is asserted right here: the same oracle with the
    # interpreter's two casts swapped in reproduces the fixture to 2e-3 on these very inputs (pins the ALGORITHM to the
    # reference's run), and the oracle with IEEE casts is what HIP is held to above.
    from oracle import fp8 as ofp8
    from tests.test_oracle_golden import interpreter_casts

    with pytest.MonkeyPatch.context() as mp:
        interpreter_casts.__wrapped__(mp)
        ref_interp = omoe.fused_experts_fp8(args[0], args[1], args[2], args[6], args[5], args[3], args[4])
    assert ofp8.CAST["fp8"].__name__ != "to_fp8_interp"  # restored
    assert max_rel_to_peak(ref_interp, bf16(g["out"])) < 2e-3, "the fixture is the oracle's algorithm under the interpreter's casts"
    assert_close(out, bf16(g["out"]), 0.2)


@pytest.mark.parametrize(
    "M,E,topk,K,I",
    [
        (1, 32, 8, 7168, 256),    # R1 TP=8 per-expert shapes, bs=1
        (16, 32, 8, 7168, 256),   # bs=16 (several tokens per expert)
        (33, 16, 4, 512, 128),    # > one 16-slot tile per expert
        (5, 8, 2, 256, 128),      # fixture shape
        (7, 8, 3, 384, 640),      # I/128 = 5 -> the wide two-launch form (quantised activations in LDS)
        (4, 64, 6, 2048, 384),    # V2-Lite-like: 64 experts, top-6
        (16, 64, 8, 2048, 1408),  # DeepSeek-V2-Lite's expert width (11 K blocks), 6 routed + 2 shared slots
        (5, 8, 2, 512, 2048),     # an expert-parallel rank's full-width R1 experts
        (40, 4, 2, 256, 1024),    # several 16-slot tiles per expert, 4 tiles per wave
    ],
)
def test_vs_oracle(M, E, topk, K, I):
    args = make_case(M, E, topk, K, I, seed=M * 1000 + E)
    out = run_hip(*args)
    x, w1, w2, w1s, w2s, ids, wts = args
    ref = omoe.fused_experts_fp8(x, w1, w2, wts, ids, w1s, w2s)
    # peak bar for every element AND the element-wise bar (rtol 1e-2 + half the peak bar) for EVERY element -- except at the one
    # shape where a census on the MI355X (tools/r06_moe_outliers.py, profiles/r06_moe_outlier_census.txt) finds any outside it:
    # R1 at bs 16, 13 of 114 688 elements, each under 0.2 % of the peak beyond its bar.  Cause: the experts' intermediate h is
    # re-quantised to fp8 between the two GEMMs, and where HIP's and the oracle's fp32 sums (different summation order over
    # K = 7168) round an h value to neighbouring fp8 codes (one step = 6 %) the 7168 outputs fed by it move; the same HIP output
    # against the reference's own kernels run on the MI355X needs no allowance at all
    # (test_fp8_experts_against_the_reference_kernels_run_on_the_mi355x).  The count is asserted, not a fraction.
    allowed = {(16, 32, 8, 7168, 256): 20}.get((M, E, topk, K, I), 0)
    assert_close(out, ref, REL_TOL, what=f"fused_experts fp8 {M, E, topk, K, I}", outlier_frac=allowed / out.numel())
    assert ((out.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item() < 5e-3


# ---------------------------------------------------------------- bf16-activation modes (fused_moe.py:232-276, :298)
def run_hip_bf16_act(x, w1, w2, ids, wts, w1s=None, w2s=None, **kw):
    from chitu_amd import fused_moe

    soft = w1s is not None
    out = fused_moe.fused_experts(
        x.cuda().clone(), w1.cuda(), w2.cuda(), wts.cuda(), ids.cuda(), inplace=False, use_fp8_w8a8=soft,
        w1_scale=w1s.cuda() if soft else None, w2_scale=w2s.cuda() if soft else None, block_shape=[128, 128],
        soft_fp8=soft, **kw)
    return out.cpu()


BF16_SHAPES = [
    (1, 32, 8, 7168, 256),    # R1 TP=8 per-expert shapes, bs 1
    (16, 32, 8, 7168, 256),   # bs 16
    (33, 16, 4, 512, 128),    # more than one 16-slot tile per expert
    (5, 8, 2, 256, 128),
    (7, 8, 3, 384, 640),
    (4, 64, 6, 2048, 1408 - 1408 % 128),  # V2-Lite-like (1280-wide here: the expert width must keep 128-blocks whole)
]


@pytest.mark.parametrize("M,E,topk,K,I", BF16_SHAPES)
def test_bf16_experts_vs_oracle(M, E, topk, K, I):
    """fused_experts(use_fp8_w8a8=False): bf16 weights, bf16 activations -- the reference's MoE for scale-less checkpoints
    (model_deepseek_v3.py:950-956).  Oracle pinned by the reference's own run (tests/golden/fused_moe_bf16.npz)."""
    g = torch.Generator().manual_seed(M * 1000 + E + 5)
    x = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * I, K, generator=g) * K ** -0.5).to(torch.bfloat16)
    w2 = (torch.randn(E, K, I, generator=g) * I ** -0.5).to(torch.bfloat16)
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)])
    wts = torch.rand(M, topk, generator=g).to(torch.bfloat16)
    out = run_hip_bf16_act(x, w1, w2, ids, wts)
    ref = omoe.fused_experts_bf16(x, w1, w2, wts, ids)
    assert_close(out, ref, REL_TOL)
    assert_close_elementwise(out, ref, what="bf16 experts")
    assert torch.equal(out, run_hip_bf16_act(x, w1, w2, ids, wts))  # deterministic


@pytest.mark.parametrize("M,E,topk,K,I", BF16_SHAPES)
def test_soft_fp8_experts_vs_oracle_and_vs_dequantised_bf16(M, E, topk, K, I):
    """fused_experts(use_fp8_w8a8=True, soft_fp8=True): fp8 weights decoded to bf16 in registers, bf16 activations
    (fused_moe.py:232-276; the README's `infer.soft_fp8=True`).  Must be the bf16 mode on weight_dequant_soft_fp8's
    output (what the reference runs on non-NVIDIA devices, model_deepseek_v3.py:975-993): same decode bit for bit,
    so only the K-block order of the fp32 sum differs."""
    from chitu_amd import ops

    x, w1, w2, w1s, w2s, ids, wts = make_case(M, E, topk, K, I, seed=M * 1000 + E + 9)
    out = run_hip_bf16_act(x, w1, w2, ids, wts, w1s, w2s)
    ref = omoe.fused_experts_soft_fp8(x, w1, w2, wts, ids, w1s, w2s)
    assert_close(out, ref, REL_TOL)
    assert_close_elementwise(out, ref, what="soft-fp8 experts")
    torch.set_default_dtype(torch.bfloat16)  # the dequant's output dtype is torch's default, as in the reference (ops.py:413)
    try:
        w1d = torch.stack([ops.weight_dequant_soft_fp8_deepseek_v3(w1[e].cuda(), w1s[e].cuda()) for e in range(E)])
        w2d = torch.stack([ops.weight_dequant_soft_fp8_deepseek_v3(w2[e].cuda(), w2s[e].cuda()) for e in range(E)])
    finally:
        torch.set_default_dtype(torch.float32)
    via_bf16 = run_hip_bf16_act(x, w1d.cpu(), w2d.cpu(), ids, wts)
    assert_close(out, via_bf16, 4e-3)# one bf16 flip of an intermediate at most


def test_bf16_act_modes_expert_map_and_unreduced_output():
    x, w1, w2, w1s, w2s, ids, wts = make_case(6, 8, 2, 256, 128, seed=19)
    emap = torch.tensor([0, 1, 2, 3, -1, -1, -1, -1], dtype=torch.int32)
    wts_masked = torch.where(ids < 4, wts.float(), torch.zeros(())).to(wts.dtype)
    out = run_hip_bf16_act(x, w1[:4].contiguous(), w2[:4].contiguous(), ids, wts, w1s[:4].contiguous(), w2s[:4].contiguous(),
                           expert_map=emap.cuda(), global_num_experts=8)
    ref = omoe.fused_experts_soft_fp8(x, w1, w2, wts_masked, ids, w1s, w2s)
    assert_close(out, ref, REL_TOL)
    un = run_hip_bf16_act(x, w1, w2, ids, wts, w1s, w2s, reduce_topk=False)
    assert tuple(un.shape) == (6, 2, 256)
    assert torch.equal(un.float().sum(1).to(torch.bfloat16), run_hip_bf16_act(x, w1, w2, ids, wts, w1s, w2s))
    from chitu_amd import fused_moe

    for bad in ({"use_int8_w8a16": True}, {"use_int4_w4a16": True}):
        with pytest.raises(NotImplementedError):
            fused_moe.fused_experts(x.cuda(), w1.cuda(), w2.cuda(), wts.cuda(), ids.cuda(), **bad)


def test_inplace_fp32_weights_and_determinism():
    args = make_case(9, 16, 4, 512, 256, seed=3, w_dtype=torch.float32)
    a = run_hip(*args, inplace=True)
    b = run_hip(*args, inplace=False)
    assert torch.equal(a, b)
    for _ in range(5):
        assert torch.equal(run_hip(*args), a)  # no atomics => bit-reproducible
    x, w1, w2, w1s, w2s, ids, wts = args
    ref = omoe.fused_experts_fp8(x, w1, w2, wts, ids, w1s, w2s)
    assert_close(a, ref, REL_TOL)


def test_expert_map_masks_remote_experts():
    """EP hook (fused_moe.py:163-179, 516-517): experts mapped to -1 contribute zeros."""
    x, w1, w2, w1s, w2s, ids, wts = make_case(6, 8, 2, 256, 128, seed=9)
    emap = torch.tensor([0, 1, 2, 3, -1, -1, -1, -1], dtype=torch.int32)
    out = run_hip(x, w1[:4].contiguous(), w2[:4].contiguous(), w1s[:4].contiguous(), w2s[:4].contiguous(),
                  ids, wts, expert_map=emap.cuda(), global_num_experts=8)
    # oracle: zero the routed weight of remote experts
    wts_masked = torch.where(ids < 4, wts.float(), torch.zeros(())).to(wts.dtype)
    ref = omoe.fused_experts_fp8(x, w1, w2, wts_masked, ids, w1s, w2s)
    assert_close(out, ref, REL_TOL)


def test_linearity_in_routed_weight_full_r1_shape():
    """Size-independent property at the R1 per-rank expert shape: doubling every routed weight
    doubles the output exactly (power-of-two scaling commutes with every rounding)."""
    x, w1, w2, w1s, w2s, ids, wts = make_case(16, 32, 8, 7168, 256, seed=21)
    a = run_hip(x, w1, w2, w1s, w2s, ids, wts)
    b = run_hip(x, w1, w2, w1s, w2s, ids, (wts.float() * 2).to(wts.dtype))
    assert torch.equal(b.float(), a.float() * 2)


@pytest.mark.parametrize("M,E,topk,K,I", [(16, 64, 8, 2048, 1408), (5, 8, 2, 512, 2048), (3, 8, 2, 256, 640)])
def test_wide_experts_with_expert_map_vs_oracle(M, E, topk, K, I):
    """Experts wider than 512 (V2-Lite, expert-parallel ranks) on a rank that holds half of them: GEMM1, SiLU + quant, generic
    GEMM2 with the other rank's slots zero-filled (fused_moe.py:40-59)."""
    x, w1, w2, w1s, w2s, ids, wts = make_case(M, E, topk, K, I, seed=M + I)
    emap = torch.full((E,), -1, dtype=torch.int32)
    emap[: E // 2] = torch.arange(E // 2, dtype=torch.int32)
    h = E // 2
    mapped = run_hip(x, w1[:h].contiguous(), w2[:h].contiguous(), w1s[:h].contiguous(), w2s[:h].contiguous(), ids, wts,
                     expert_map=emap.cuda(), global_num_experts=E)
    ref = omoe.fused_experts_fp8(x, w1, w2, torch.where(ids < h, wts.float(), torch.zeros(())).to(wts.dtype), ids, w1s, w2s)
    assert_close(mapped, ref, REL_TOL, atol_frac=1.0)


@pytest.mark.parametrize("M,E,topk,K,I", [(16, 64, 8, 2048, 1408), (64, 8, 2, 512, 2048), (40, 32, 6, 1024, 640)])
def test_wide_experts_three_launch_path_vs_oracle(M, E, topk, K, I):
    """Wide experts (I > 512: V2-Lite's 1408, EP ranks' full-width experts) take gemm1 + silu_mul_quant + gemm2; expert_map too."""
    args = make_case(M, E, topk, K, I, seed=300 + M)
    x, w1, w2, w1s, w2s, ids, wts = args
    assert_close(run_hip(*args), omoe.fused_experts_fp8(x, w1, w2, wts, ids, w1s, w2s), REL_TOL)


@pytest.mark.parametrize("M,E,topk,K,I", [(1, 32, 8, 7168, 256), (16, 32, 8, 7168, 256), (33, 16, 4, 512, 128), (4, 64, 6, 2048, 384),
                                          (3, 8, 2, 256, 512), (70, 8, 4, 1024, 256)])
def test_two_launch_expert_path_is_bit_identical_to_three_launch(M, E, topk, K, I, monkeypatch):
    """gemm1_silu + gemm2_quant == gemm1 + silu_mul_quant + gemm2 (all K-split variants, expert_map too)."""
    args = make_case(M, E, topk, K, I, seed=77 + M)
    emap = torch.arange(E, dtype=torch.int32)
    emap[E // 2:] = -1
    # the two forms pick their K split (waves per workgroup) from different launch heuristics; with the
    # same split the fp32 summation order, and therefore every bit, is the same
    from chitu_amd._lib import debug_option

    for wk in (1, 2, 8):
        with debug_option("moe_gemm1_wk", wk):
            outs = {}
            for fuse in ("1", "0"):
                monkeypatch.setenv("CHITU_MOE_FUSE_SILU", fuse)
                outs[fuse] = (run_hip(*args), run_hip(*args, expert_map=emap.cuda(), global_num_experts=E))
        assert torch.equal(outs["1"][0], outs["0"][0]), wk
        assert torch.equal(outs["1"][1], outs["0"][1]), wk
    # default heuristics: same function up to the summation order
    monkeypatch.setenv("CHITU_MOE_FUSE_SILU", "1")
    a = run_hip(*args)
    monkeypatch.setenv("CHITU_MOE_FUSE_SILU", "0")
    assert_close(a, run_hip(*args), 4e-3)


@pytest.mark.parametrize("M,E,topk,K,I", [(128, 32, 8, 7168, 256), (300, 16, 4, 512, 128), (1000, 64, 6, 2048, 384),
                                          (2048, 64, 8, 7168, 256), (129, 8, 2, 256, 640), (300, 8, 2, 128, 128)])
def test_prefill_tiled_expert_path_vs_decode_kernels_and_oracle(M, E, topk, K, I, monkeypatch):
    """>= 128 tokens: moe_align with block 64 + the tiled grouped GEMMs (csrc/moe_tiled.hip) against the decode kernels on
    the same inputs (tiling switched off; the last case is ONE K block per GEMM: a ring that never turns): same rounding
    points, only the order of the sum inside a 128-block differs
    (the bf16 outputs agree to a last bit here and there: <= 2e-2 of the peak, mean difference < 1e-3); against the CPU
    oracle on the small cases <= 1e-2; run to run identical."""
    from chitu_amd import fused_moe

    args = make_case(M, E, topk, K, I, seed=M + E)
    tiled = run_hip(*args)
    assert torch.equal(tiled, run_hip(*args))
    monkeypatch.setattr(fused_moe, "_MOE_TILED_MIN_TOKENS", 0)
    streamed = run_hip(*args)
    # a bf16 output's last bit is up to 2^-7 of the peak, and an output is the rounded sum of topk such values
    worst = max_rel_to_peak(tiled, streamed)
    mean = ((tiled.float() - streamed.float()).abs().mean() / streamed.float().abs().mean()).item()
    assert worst < 2e-2 and mean < 1e-3, (worst, mean)
    if M * topk * K * I <= 3e9:
        x, w1, w2, w1s, w2s, ids, wts = args
        ref = omoe.fused_experts_fp8(x, w1, w2, wts, ids, w1s, w2s)
        # hundreds of rows: the largest single deviation (one h value on an fp8 rounding boundary, SiLU by expf vs
        # torch.exp) grows with the element count and is shared with the decode kernels; the mean stays put
        assert_close(tiled, ref, 2e-2)
        assert max_rel_to_peak(streamed, ref) < 2e-2
        assert ((tiled.float() - ref.float()).abs().mean() / ref.float().abs().mean()).item() < 5e-3


def test_prefill_tiled_expert_path_with_expert_map():
    """Expert parallelism at prefill size: slots of experts mapped to -1 come out as zeros of the tiled kernels too."""
    x, w1, w2, w1s, w2s, ids, wts = make_case(200, 8, 2, 256, 128, seed=19)
    emap = torch.tensor([0, 1, 2, 3, -1, -1, -1, -1], dtype=torch.int32)
    out = run_hip(x, w1[:4].contiguous(), w2[:4].contiguous(), w1s[:4].contiguous(), w2s[:4].contiguous(),
                  ids, wts, expert_map=emap.cuda(), global_num_experts=8)
    wts_masked = torch.where(ids < 4, wts.float(), torch.zeros(())).to(wts.dtype)
    ref = omoe.fused_experts_fp8(x, w1, w2, wts_masked, ids, w1s, w2s)
    assert_close(out, ref, REL_TOL, atol_frac=1.0)  # 64-slot tiles: near-zero outputs carry their tile's rounding


@pytest.mark.parametrize("case", ["r1_bs16", "r1_bs1", "small"])
def test_fp8_experts_against_the_reference_kernels_run_on_the_mi355x(case):
    """fused_experts_impl(use_fp8_w8a8, block_shape [128, 128]) vs tests/golden/hw_fused_moe_fp8.npz: the reference's
    fused_moe_kernel + moe_align stages + per_token_group_quant_fp8 + SiluAndMul (fused_moe.py:62-307, 314-442, 640-720,
    24-39) compiled by Triton-ROCm and run on an MI355X on the same seeded inputs (R1's per-rank expert shapes, 32
    experts).  Direct bar: the north_star's 1e-2 of the peak, element-wise form included, no outlier allowance."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import hw_cases as hc
    from chitu_amd import fused_moe

    g = golden("hw_fused_moe_fp8")
    x, w1, w2, w1s, w2s, ids, wts = hc.fused_moe_fp8_case(case)
    out = fused_moe.fused_experts_impl(x.cuda(), w1.cuda(), w2.cuda(), wts.cuda(), ids.cuda(), inplace=False, use_fp8_w8a8=True,
                                       w1_scale=w1s.cuda(), w2_scale=w2s.cuda(), block_shape=[128, 128])
    assert_close(out, bf16(g[f"{case}_out"]), 1e-2, what=case)


@pytest.mark.parametrize("case", ["bs16", "small"])
def test_bf16_experts_against_the_reference_kernels_run_on_the_mi355x(case):
    """fused_experts_impl(use_fp8_w8a8=False) on bf16 weights vs tests/golden/hw_fused_moe_bf16.npz (the reference's bf16
    tl.dot path, which the Triton interpreter cannot run at all: SURVEY 8c)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import hw_cases as hc
    from chitu_amd import fused_moe

    g = golden("hw_fused_moe_bf16")
    x, w1, w2, ids, wts = hc.fused_moe_bf16_case(case)
    out = fused_moe.fused_experts_impl(x.cuda(), w1.cuda(), w2.cuda(), wts.cuda(), ids.cuda(), inplace=False, use_fp8_w8a8=False)
    assert_close(out, bf16(g[f"{case}_out"]), 5e-3, what=case)


@pytest.mark.parametrize("case", ["r1_bs16", "small"])
def test_soft_fp8_moe_branch_against_the_reference_kernels_run_on_the_mi355x(case):
    """The README configuration's MoE branch off NVIDIA (infer.soft_fp8=True, model_deepseek_v3.py:975-996):
    weight_dequant_soft_fp8_deepseek_v3 over the stacked experts, then fused_experts(use_fp8_w8a8=False), vs
    tests/golden/hw_soft_fp8_moe.npz -- the reference's two soft-dequant Triton kernels (ops.py:396-449) + its bf16 fused MoE
    compiled by Triton-ROCm and run on an MI355X on the same seeded inputs.  The dequantised tensor is bit-exact."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import hw_cases as hc
    from chitu_amd import fused_moe, ops

    g = golden("hw_soft_fp8_moe")
    x, w1, w2, w1s, w2s, ids, wts = hc.soft_fp8_moe_case(case)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        w1d = ops.weight_dequant_soft_fp8_deepseek_v3(w1.cuda(), w1s.cuda(), 128)
        w2d = ops.weight_dequant_soft_fp8_deepseek_v3(w2.cuda(), w2s.cuda(), 128)
    finally:
        torch.set_default_dtype(prev)
    if case == "small":
        assert np.array_equal(hc.bits16(w1d), g["small_w1_dequant"])
    out = fused_moe.fused_experts(x.cuda(), w1d, w2d, wts.cuda(), ids.cuda(), inplace=False, use_fp8_w8a8=False)
    assert_close(out, bf16(g[f"{case}_out"]), 5e-3, what=case)
