"""Host tests: the oracle (IEEE casts) against the HARDWARE goldens -- the reference's own Triton kernels compiled by
Triton-ROCm and run on an MI355X (tests/golden/gen_hw_golden.py; inputs rebuilt from seeds by tests/golden/hw_cases.py).
This is the direct link oracle <-> reference execution on real casts: no Triton-interpreter cast defects in between
(the CPU fixtures of tests/golden/gen_golden.py carry those, and the oracle reproduces them only in its emulation mode).
Outcome, asserted below: act_quant is BIT-EXACT; every GEMM-shaped result agrees to the last bf16 bit on 97-100 % of its
elements and within one bf16 ulp of the tensor's peak on the rest -- the fp32 summation order inside a 128-wide K block
(tl.dot's MFMA tree vs numpy's pairwise sum), nothing else."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import hw_cases as hc  # noqa: E402

from oracle import fp8 as ofp8  # noqa: E402
from oracle import gqa as ogqa  # noqa: E402
from oracle import mla as omla  # noqa: E402
from oracle import moe as omoe  # noqa: E402
from tests.util import bf16, golden, max_rel_to_peak  # noqa: E402


def _agree(out, want_bits, peak_tol, min_equal_frac, what):
    want = bf16(want_bits)
    got = out.to(torch.bfloat16)
    assert tuple(got.shape) == tuple(want.shape), what
    err = max_rel_to_peak(got, want)
    equal = float((hc.bits16(got) == want_bits).mean())
    assert err < peak_tol and equal >= min_equal_frac, (what, err, equal)
    return err, equal


@pytest.mark.parametrize("case", hc.FP8_LINEAR_CASES)
def test_act_quant_is_bit_exact_and_fp8_gemm_within_one_ulp_of_the_reference_on_hardware(case):
    g = golden("hw_fp8_linear")
    x, w, ws = hc.fp8_linear_case(case)
    xq, xs = ofp8.act_quant_deepseek_v3(x)
    assert np.array_equal(hc.bits8(xq), g[f"{case}_xq"]) and np.array_equal(xs.numpy(), g[f"{case}_xs"])  # K5: bit for bit
    c = ofp8.fp8_gemm_deepseek_v3(xq, xs, w, ws, torch.bfloat16)
    _agree(c, g[f"{case}_c"], 4e-3, 0.97, case)  # one bf16 ulp at the peak binade = 3.9e-3
    if case == "ragged":  # K8 dequant: exact products of an fp8 value and an fp32 scale, rounded once
        wd = ofp8.weight_dequant_deepseek_v3(w, ws)
        assert np.array_equal(hc.bits16(wd), g["ragged_w_dequant"])


@pytest.mark.parametrize("case", hc.FUSED_MOE_FP8_CASES)
def test_fused_experts_fp8_oracle_vs_the_reference_on_hardware(case):
    g = golden("hw_fused_moe_fp8")
    x, w1, w2, w1s, w2s, ids, wts = hc.fused_moe_fp8_case(case)
    o = omoe.fused_experts_fp8(x, w1, w2, wts, ids, w1s, w2s)
    # two fp8 re-quantisations inside (fused_moe.py:829): a summation-order ulp in GEMM1 can flip an fp8 code of h
    _agree(o, g[f"{case}_out"], 5e-3, 0.88, case)


@pytest.mark.parametrize("case", hc.FUSED_MOE_BF16_CASES)
def test_fused_experts_bf16_oracle_vs_the_reference_on_hardware(case):
    g = golden("hw_fused_moe_bf16")
    x, w1, w2, ids, wts = hc.fused_moe_bf16_case(case)
    o = omoe.fused_experts_bf16(x, w1, w2, wts, ids)
    _agree(o, g[f"{case}_out"], 2e-3, 0.999, case)


@pytest.mark.parametrize("case", hc.MLA_DECODE_CASES)
def test_mla_decode_oracle_vs_the_reference_on_hardware(case):
    g = golden("hw_mla_decode")
    cache, qn, qp, table, lens, scale = hc.mla_decode_case(case)
    o = omla.mla_decode(qn, qp, cache, table, lens, scale)
    # the reference splits the keys in 4 and rounds P to bf16 per 64-key tile against its running maximum: bf16-ulp noise
    # (one ulp of a bf16 output is up to 7.8e-3 of the peak when the peak sits low in its binade)
    _agree(o, g[f"{case}_out"], 8e-3, 0.65, case)


@pytest.mark.parametrize("case", hc.SOFT_FP8_MOE_CASES)
def test_soft_fp8_moe_branch_oracle_vs_the_reference_on_hardware(case):
    """oracle/fp8.py's soft-fp8 dequant (bit placement, ops.py:396-449) + oracle/moe.py's bf16 fused experts vs the reference's
    kernels run on the MI355X (hw_soft_fp8_moe.npz): the dequantised experts bit-exact, the MoE output within one bf16 ulp of
    the peak."""
    g = golden("hw_soft_fp8_moe")
    x, w1, w2, w1s, w2s, ids, wts = hc.soft_fp8_moe_case(case)
    w1d = ofp8.weight_dequant_soft_fp8_deepseek_v3(w1, w1s)
    w2d = ofp8.weight_dequant_soft_fp8_deepseek_v3(w2, w2s)
    if case == "small":
        assert np.array_equal(hc.bits16(w1d.to(torch.bfloat16)), g["small_w1_dequant"])
    out = omoe.fused_experts_bf16(x, w1d.to(torch.bfloat16), w2d.to(torch.bfloat16), wts, ids)
    assert max_rel_to_peak(out, hc.from_bits16(g[f"{case}_out"])) < 5e-3


@pytest.mark.parametrize("case", hc.GQA_DECODE_CASES)
def test_gqa_decode_oracle_vs_the_reference_attention_on_hardware(case):
    g = golden("hw_gqa")
    q, kc, vc, kn, vn, lens = hc.gqa_decode_case(case)
    # the oracle's paged form with one page per sequence = the contiguous caches the reference's pure-torch path takes
    table = torch.arange(q.shape[0], dtype=torch.int32).view(-1, 1)
    ref, _, _ = ogqa.attn_with_kvcache(q, kc, vc, kn, vn, lens, table, softmax_scale=128 ** -0.5)
    assert max_rel_to_peak(ref, hc.from_bits16(g[f"decode_{case}_out"])) < 5e-3


@pytest.mark.parametrize("case", hc.GQA_PREFILL_CASES)
def test_gqa_prefill_oracle_vs_the_reference_attention_on_hardware(case):
    g = golden("hw_gqa")
    q, k, v, cu, seqs = hc.gqa_prefill_case(case)
    rows = torch.from_numpy(hc.gqa_prefill_rows(seqs))
    assert max_rel_to_peak(ogqa.attn_varlen_causal(q, k, v, cu)[rows], hc.from_bits16(g[f"prefill_{case}_out"])) < 5e-3
