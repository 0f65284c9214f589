"""HIP fp8 quant / dequant / GEMM vs the oracle and the reference fixtures."""

import numpy as np
import pytest
import torch

from oracle import fp8 as ofp8
from tests.util import assert_close, bf16, bits16, bits8, fp8, golden, max_rel_to_peak

pytestmark = pytest.mark.gpu

REL_TOL = 1e-2  # "<= 1e-2 rel for fp8 GEMM"


def randw(n, k, g):
    w = (torch.randn(n, k, generator=g, dtype=torch.float32) * 0.5).to(torch.float8_e4m3fn)
    s = torch.rand((n + 127) // 128, k // 128, generator=g, dtype=torch.float32) * 0.02 + 0.01
    return w, s


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(1, 128), (5, 512), (16, 7168), (3, 2, 256)])
def test_act_quant_bit_exact(shape, dtype):
    from chitu_amd import ops

    g = torch.Generator().manual_seed(sum(shape))
    x = (torch.randn(*shape, generator=g) * 1.7).to(dtype)
    q_ref, s_ref = ofp8.act_quant_deepseek_v3(x)
    q, s = ops.act_quant_deepseek_v3(x.cuda())
    assert q.dtype == torch.float8_e4m3fn and s.shape == s_ref.shape
    assert np.array_equal(s.cpu().numpy(), s_ref.numpy())
    assert np.array_equal(bits8(q), bits8(q_ref))


def test_act_quant_fixture_scales_and_zero_group():
    from chitu_amd import fused_moe, ops

    g = golden("fp8_linear")
    q, s = ops.act_quant_deepseek_v3(bf16(g["x"]).cuda())
    assert np.array_equal(s.cpu().numpy(), g["xs"])
    # all-zero group: reference act_quant gives scale 0 and NaN codes (0/0)
    z = torch.zeros(2, 256, dtype=torch.bfloat16)
    z[1, 128:] = 1.0
    q, s = ops.act_quant_deepseek_v3(z.cuda())
    assert s.cpu()[0, 0] == 0 and torch.isnan(q.cpu().float()[0, :128]).all()
    assert (q.cpu().float()[1, 128:] == 448).all()
    # per_token_group_quant (eps + clamp): zero group stays zero, scale eps/448
    q2, s2 = fused_moe.per_token_group_quant_fp8(z.cuda(), 128)
    q2r, s2r = ofp8.per_token_group_quant_fp8(z)
    assert np.array_equal(bits8(q2), bits8(q2r)) and np.array_equal(s2.cpu().numpy(), s2r.numpy())
    gq = golden("group_quant")
    q3, s3 = fused_moe.per_token_group_quant_fp8(bf16(gq["x"]).cuda(), 128)
    q3r, s3r = ofp8.per_token_group_quant_fp8(bf16(gq["x"]))
    assert np.array_equal(s3.cpu().numpy(), gq["s"]) and np.array_equal(bits8(q3), bits8(q3r))


def test_weight_dequant_all_codes_and_shapes():
    from chitu_amd import ops

    torch.set_default_dtype(torch.bfloat16)
    try:
        codes = torch.arange(256, dtype=torch.uint8).repeat(128).reshape(128, 256).view(torch.float8_e4m3fn)
        s = torch.tensor([[0.0173, 0.0291]], dtype=torch.float32)
        hard = ops.weight_dequant_deepseek_v3(codes.cuda(), s.cuda()).cpu()
        soft = ops.weight_dequant_soft_fp8_deepseek_v3(codes.cuda(), s.cuda()).cpu()
        hard_ref = ofp8.weight_dequant_deepseek_v3(codes, s)
        soft_ref = ofp8.weight_dequant_soft_fp8_deepseek_v3(codes, s)
        finite = ~torch.isnan(hard_ref.float())
        assert torch.equal(hard[finite], hard_ref[finite]) and torch.isnan(hard.float()[~finite]).all()
        assert np.array_equal(bits16(soft), bits16(soft_ref))  # incl. NaN codes -> +-480*s
        g = torch.Generator().manual_seed(3)
        w, ws = randw(384, 512, g)  # fixture shape
        gf = golden("fp8_linear")
        wd = ops.weight_dequant_deepseek_v3(fp8(gf["w"]).cuda(), torch.from_numpy(gf["ws"]).cuda())
        assert np.array_equal(bits16(wd), bits16(ofp8.weight_dequant_deepseek_v3(fp8(gf["w"]), torch.from_numpy(gf["ws"]))))
        # 3-D (stacked experts) and ragged (rows/cols not multiples of 128 / 16)
        w3 = (torch.randn(3, 200, 136, generator=g, dtype=torch.float32) * 0.5).to(torch.float8_e4m3fn)
        s3 = torch.rand(3, 2, 2, generator=g, dtype=torch.float32) * 0.02 + 0.01
        for fn, ref in ((ops.weight_dequant_deepseek_v3, ofp8.weight_dequant_deepseek_v3),
                        (ops.weight_dequant_soft_fp8_deepseek_v3, ofp8.weight_dequant_soft_fp8_deepseek_v3)):
            assert np.array_equal(bits16(fn(w3.cuda(), s3.cuda())), bits16(ref(w3, s3)))
    finally:
        torch.set_default_dtype(torch.float32)


def test_fp8_gemm_fixture():
    from chitu_amd import ops

    g = golden("fp8_linear")
    torch.set_default_dtype(torch.bfloat16)
    try:
        c = ops.fp8_gemm_deepseek_v3(
            fp8(g["xq"]).cuda(), torch.from_numpy(g["xs"]).cuda(), fp8(g["w"]).cuda(), torch.from_numpy(g["ws"]).cuda()
        )
    finally:
        torch.set_default_dtype(torch.float32)
    assert c.dtype == torch.bfloat16 and tuple(c.shape) == (5, 384)
    # fixture was truncated to bf16 by the interpreter: allow one bf16 ulp per element
    assert_close(c, bf16(g["c"]), 8e-3)
    exact = ofp8.fp8_gemm_deepseek_v3(fp8(g["xq"]), torch.from_numpy(g["xs"]), fp8(g["w"]), torch.from_numpy(g["ws"]), torch.float32)
    assert_close(c, exact, 4e-3)


R1_SHAPES = [(2112, 7168), (3072, 1536), (7168, 2048), (512, 7168), (7168, 256), (4608, 7168), (7168, 2304)]


@pytest.mark.parametrize("N,K", R1_SHAPES + [(24, 128), (1000, 384), (129, 256)])
@pytest.mark.parametrize("M", [1, 3, 16, 17, 32, 50, 70])
def test_fp8_gemm_vs_oracle(M, N, K):
    from chitu_amd import ops

    if M > 32 and N * K > 8e6:
        pytest.skip("large-M cases only on small shapes")
    g = torch.Generator().manual_seed(M * 131 + N + K)
    x = (torch.randn(M, K, generator=g) * 0.8).to(torch.bfloat16)
    w, ws = randw(N, K, g)
    xq, xs = ofp8.act_quant_deepseek_v3(x)
    ref = ofp8.fp8_gemm_deepseek_v3(xq, xs, w, ws, torch.float32)
    torch.set_default_dtype(torch.bfloat16)
    try:
        c = ops.fp8_gemm_deepseek_v3(xq.cuda(), xs.cuda(), w.cuda(), ws.cuda())
    finally:
        torch.set_default_dtype(torch.float32)
    assert tuple(c.shape) == (M, N)
    assert_close(c, ref, 5e-3)# bf16 output rounding only
    assert_close(c, ref, REL_TOL)
    # deterministic (no atomics in the split-K path)
    torch.set_default_dtype(torch.bfloat16)
    try:
        c2 = ops.fp8_gemm_deepseek_v3(xq.cuda(), xs.cuda(), w.cuda(), ws.cuda())
    finally:
        torch.set_default_dtype(torch.float32)
    assert torch.equal(c, c2)


@pytest.mark.parametrize("N,K", [(2112, 7168), (7168, 256), (136, 384)])
@pytest.mark.parametrize("M", [1, 16, 20])
def test_soft_fp8_gemm_vs_oracle(M, N, K):
    from chitu_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 0.8).to(torch.bfloat16)
    w, ws = randw(N, K, g)
    ref = ofp8.soft_fp8_gemm_deepseek_v3(x, w, ws, torch.float32)
    torch.set_default_dtype(torch.bfloat16)
    try:
        c = ops.soft_fp8_gemm_deepseek_v3(x.cuda(), w.cuda(), ws.cuda())
    finally:
        torch.set_default_dtype(torch.float32)
    assert_close(c, ref, 5e-3)


def test_linearity_property_full_size():
    """Size-independent property at the R1 wqkv_a size: gemm(a, W) + gemm(b, W) == gemm over
    stacked rows, and a row scaled by 2 (exact in fp8 scales) doubles the output exactly."""
    from chitu_amd import ops

    g = torch.Generator().manual_seed(11)
    N, K = 2112, 7168
    x = (torch.randn(4, K, generator=g)).to(torch.bfloat16)
    w, ws = randw(N, K, g)
    torch.set_default_dtype(torch.bfloat16)
    try:
        xq, xs = ops.act_quant_deepseek_v3(x.cuda())
        full = ops.fp8_gemm_deepseek_v3(xq, xs, w.cuda(), ws.cuda())
        rows = [ops.fp8_gemm_deepseek_v3(xq[i : i + 1].contiguous(), xs[i : i + 1].contiguous(), w.cuda(), ws.cuda()) for i in range(4)]
        assert torch.equal(full, torch.cat(rows))  # row results do not depend on the batch
        dbl = ops.fp8_gemm_deepseek_v3(xq, (xs * 2).contiguous(), w.cuda(), ws.cuda())
        assert torch.equal(dbl.float(), full.float() * 2)
    finally:
        torch.set_default_dtype(torch.float32)


def test_arithmetic_shortcuts_selftest():
    """The quantising kernels' group division (refined reciprocal + residual correction) and the
    hardware bf16 rounding against their slow definitions, on the device, over 48M operand pairs:
    quantisation-shaped (|x| <= amax, d = amax/448), bf16-grid operands (where exact ties of the
    e4m3 rounding live) and wide-range random floats."""
    import ctypes

    from chitu_amd import _lib
    from chitu_amd._lib import i64, ptr, stream_ptr

    g = torch.Generator(device="cuda").manual_seed(11)
    n = 1 << 24
    bad = torch.zeros(3, dtype=torch.int64, device="cuda")
    cases = []
    amax = torch.rand(n, device="cuda", generator=g) * 8 + 1e-3
    cases.append(((torch.rand(n, device="cuda", generator=g) * 2 - 1) * amax, amax / 448.0))
    xb = (torch.randn(n, device="cuda", generator=g) * 3).to(torch.bfloat16).float()
    ab = (torch.rand(n, device="cuda", generator=g) * 6 + 0.01).to(torch.bfloat16).float()
    cases.append((xb, ab / 448.0))
    # wide dynamic range, quotient kept inside the normal range (the kernels' |x| <= amax = 448 d)
    ed = torch.randint(-55, 55, (n,), device="cuda", generator=g).float()
    ex = ed + torch.randint(-30, 30, (n,), device="cuda", generator=g).float()
    cases.append((torch.randn(n, device="cuda", generator=g) * torch.exp2(ex),
                  (torch.rand(n, device="cuda", generator=g) + 0.5) * torch.exp2(ed)))
    for k, (num, den) in enumerate(cases):
        bad.zero_()
        rc = _lib.lib().chitu_hip_selftest_arith(ptr(num.contiguous()), ptr(den.contiguous()), i64(n), ptr(bad), stream_ptr())
        assert rc == 0
        torch.cuda.synchronize()
        f32_mismatch, fp8_mismatch, bf16_mismatch = bad.tolist()
        assert fp8_mismatch == 0 and bf16_mismatch == 0 and f32_mismatch == 0, (k, bad.tolist())


@pytest.mark.parametrize("M,N,K,S", [(1, 2112, 7168, 2), (16, 2112, 7168, 2), (32, 2112, 7168, 2), (20, 512, 7168, 4),
                                     (5, 1000, 384, 3), (70, 136, 512, 2)])
def test_fp8_gemm_split_k_partial_planes(M, N, K, S):
    """chitu_hip_fp8_gemm_blockscale_partials: plane s holds the contribution of its K blocks, the planes' fp32 sum
    is the GEMM (vs the oracle <= 5e-3 after the bf16 rounding the consumer applies), run to run identical."""
    from chitu_amd import ops

    g = torch.Generator().manual_seed(M * 17 + N + K + S)
    x = (torch.randn(M, K, generator=g) * 0.8).to(torch.bfloat16)
    w, ws = randw(N, K, g)
    xq, xs = ofp8.act_quant_deepseek_v3(x)
    ref = ofp8.fp8_gemm_deepseek_v3(xq, xs, w, ws, torch.float32)
    parts = ops.fp8_gemm_partials_deepseek_v3(xq.cuda(), xs.cuda(), w.cuda(), ws.cuda(), S)
    assert tuple(parts.shape) == (S, M, N) and parts.dtype == torch.float32
    total = parts[0].clone()
    for s in range(1, S):
        total += parts[s]
    assert_close(total.to(torch.bfloat16), ref, 5e-3)
    # each plane is a partial contraction over a contiguous K range: plane s == the GEMM on that range alone
    KB = K // 128
    T = S * (8 if KB >= 8 * S else 4 if KB >= 4 * S else 2 if KB >= 2 * S else 1)
    for s in range(S):
        k0, k1 = (KB * (s * (T // S)) // T) * 128, (KB * ((s + 1) * (T // S)) // T) * 128
        ref_s = ofp8.fp8_gemm_deepseek_v3(xq[:, k0:k1].contiguous(), xs[:, k0 // 128:k1 // 128].contiguous(),
                                          w[:, k0:k1].contiguous(), ws[:, k0 // 128:k1 // 128].contiguous(), torch.float32)
        assert_close(parts[s], ref_s, 1e-4, what=s)
    assert torch.equal(parts, ops.fp8_gemm_partials_deepseek_v3(xq.cuda(), xs.cuda(), w.cuda(), ws.cuda(), S))


@pytest.mark.parametrize("M,N,K", [(128, 2112, 7168), (2048, 3072, 1536), (1000, 7168, 2048), (129, 200, 128), (333, 136, 384),
                                   (4096, 512, 7168)])
def test_fp8_gemm_tiled_prefill_form_vs_streaming_form_and_oracle(M, N, K):
    """M >= 128 takes the compute-shaped kernel (fp8_gemm_tiled.hip: 128 x 128 tiles through LDS).  Against the
    weight-streaming kernel forced on the same inputs (launch-variant option fp8_gemm_tiled = 0): fp32 outputs agree to
    summation order inside a 128-block (<= 1e-5 of the peak); against the CPU oracle on the small shapes: <= 5e-3 after
    the bf16 rounding.  Ragged edges (M, N not multiples of 128 / 16) included; run to run identical."""
    from chitu_amd import _lib, ops

    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) * 0.8).to(torch.bfloat16)
    w, ws = randw(N, K, g)
    xq, xs = ofp8.act_quant_deepseek_v3(x)
    xq, xs, w, ws = xq.cuda(), xs.cuda(), w.cuda(), ws.cuda()
    tiled = ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.float32)
    with _lib.debug_option("fp8_gemm_tiled", 0):
        streamed = ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.float32)
    assert tuple(tiled.shape) == (M, N) and torch.isfinite(tiled).all()
    assert_close(tiled, streamed, 1e-5)
    assert torch.equal(tiled, ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.float32))
    b16 = ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.bfloat16)
    assert_close(b16, streamed, 5e-3)
    if M * N * K <= 2e8:
        ref = ofp8.fp8_gemm_deepseek_v3(xq.cpu(), xs.cpu(), w.cpu(), ws.cpu(), torch.float32)
        assert_close(b16, ref, 5e-3)


@pytest.mark.parametrize("M,N,K", [(128, 2112, 7168), (2048, 2112, 7168), (1000, 7168, 2048), (129, 200, 128), (333, 136, 384), (65, 130, 256)])
def test_fp8_gemm_tiled_token_tile_heights_return_the_same_bits(M, N, K):
    """Round 6: 64-token tiles for grids that would leave CUs with fewer than two workgroups (launcher heuristic; option
    fp8_tiled_tm forces).  The arithmetic per output element and its order are the 128-token form's: bit-identical fp32 and
    bf16 outputs on full, ragged and single-tile shapes."""
    from chitu_amd import _lib, ops

    g = torch.Generator().manual_seed(M * 3 + N + K)
    x = (torch.randn(M, K, generator=g) * 0.8).to(torch.bfloat16)
    w, ws = randw(N, K, g)
    xq, xs = ofp8.act_quant_deepseek_v3(x)
    xq, xs, w, ws = xq.cuda(), xs.cuda(), w.cuda(), ws.cuda()
    outs = {}
    for tm in (64, 128):
        with _lib.debug_option("fp8_tiled_tm", tm):
            outs[tm] = (ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.float32),
                        ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.bfloat16))
    assert torch.isfinite(outs[64][0]).all()
    assert torch.equal(outs[64][0], outs[128][0]) and torch.equal(outs[64][1], outs[128][1])
    assert torch.equal(outs[64][0], ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.float32))  # whichever the heuristic picks


# ---------------------------------------------------------------- tile-major activations (the fused step's internal layout)
@pytest.mark.parametrize("M", [1, 5, 16, 17, 32, 33, 64])
@pytest.mark.parametrize("N,K", [(2112, 7168), (7168, 2048), (3072, 1536), (200, 512)])
def test_tile_major_activations_are_the_same_gemm_bit_for_bit(M, N, K):
    """ops.TiledQuant: the fp8 activations laid out [tile][K/16][16 rows][16 B] (+ scales [tile][K/128][16]) as the
    fused decode step keeps them between the norm / W_UV launches and the GEMMs that read them.  Only the addresses
    change: (1) rms_norm(add=, quant=, tile_major=True) writes the SAME codes and scales as the row-major form, permuted;
    (2) the GEMM on them returns the SAME bits as the row-major GEMM -- every K-split variant, ragged M and N."""
    from chitu_amd import _lib, ops

    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    add = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    wn = (1 + 0.1 * torch.randn(K, generator=g)).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g) * 0.5).to(torch.float8_e4m3fn).cuda()
    ws = (torch.rand((N + 127) // 128, K // 128, generator=g) * 0.02 + 0.01).cuda()
    for quant in ("act", "group"):
        x1, y1, q1, s1 = ops.rms_norm(x, wn, 1e-6, quant=quant, add=add)
        x2, y2, tq, none = ops.rms_norm(x, wn, 1e-6, quant=quant, add=add, tile_major=True)
        assert none is None and isinstance(tq, ops.TiledQuant) and torch.equal(x1, x2) and torch.equal(y1, y2)
        q2, s2 = tq.to_row_major()
        assert torch.equal(q1.view(torch.uint8), q2.view(torch.uint8)) and torch.equal(s1, s2)
        ref = ops.fp8_gemm_deepseek_v3(q1, s1, w, ws, out_dtype=torch.bfloat16)
        out = ops.fp8_gemm_deepseek_v3(tq, None, w, ws, out_dtype=torch.bfloat16)
        assert torch.equal(out, ref)
        for wk in (1, 2, 4, 8):
            with _lib.debug_option("fp8_gemm_wk", wk):
                assert torch.equal(ops.fp8_gemm_deepseek_v3(tq, None, w, ws, out_dtype=torch.float32),
                                   ops.fp8_gemm_deepseek_v3(q1, s1, w, ws, out_dtype=torch.float32)), wk


@pytest.mark.parametrize("case", ["wqkv_a", "wo", "wq_b_m64", "ragged"])
def test_against_the_reference_kernels_run_on_the_mi355x(case):
    """HIP act_quant + fp8_gemm vs tests/golden/hw_fp8_linear.npz: the reference's own Triton kernels (ops.py:330-353,
    453-483) compiled by Triton-ROCm and run on an MI355X (tests/golden/gen_hw_golden.py) on the same seeded inputs, R1
    per-rank shapes.  Real RTNE casts on both sides: the quantisation is bit for bit, the GEMM within the north_star's
    1e-2 directly (measured ~3e-3 = one bf16 ulp of the peak: fp32 summation order inside a K block)."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import hw_cases as hc
    from chitu_amd import ops

    g = golden("hw_fp8_linear")
    x, w, ws = hc.fp8_linear_case(case)
    xq, xs = ops.act_quant_deepseek_v3(x.cuda())
    assert np.array_equal(bits8(xq), g[f"{case}_xq"]) and np.array_equal(xs.cpu().numpy(), g[f"{case}_xs"])
    c = ops.fp8_gemm_deepseek_v3(xq, xs, w.cuda(), ws.cuda(), out_dtype=torch.bfloat16)
    assert_close(c, bf16(g[f"{case}_c"]), 5e-3, what=case)
    if case == "ragged":
        torch.set_default_dtype(torch.bfloat16)  # the reference's output dtype rule (ops.py:357-392)
        try:
            wd = ops.weight_dequant_deepseek_v3(w.cuda(), ws.cuda())
        finally:
            torch.set_default_dtype(torch.float32)
        assert wd.dtype == torch.bfloat16 and np.array_equal(bits16(wd), g["ragged_w_dequant"])
