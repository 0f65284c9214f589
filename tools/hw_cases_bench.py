#!/usr/bin/env python3
"""chitu_amd's ops on the cases of tests/golden/hw_cases.py, timed the way tests/golden/gen_hw_golden.py times the
reference's Triton kernels (HIP events over 20 calls issued from Python, host launch path included):
python tools/hw_cases_bench.py >> profiles/r05_reference_triton_on_mi355x.txt"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import hw_cases as hc  # noqa: E402


def time_us(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


@torch.inference_mode()
def main():
    from chitu_amd import fused_moe, ops
    from chitu_amd.attn_backend import HipAttnBackend

    rows = []
    for case in hc.FP8_LINEAR_CASES:
        x, w, ws = [t.cuda() for t in hc.fp8_linear_case(case)]
        xq, xs = ops.act_quant_deepseek_v3(x)
        rows.append(("chitu_amd ops.act_quant_deepseek_v3", case, time_us(lambda: ops.act_quant_deepseek_v3(x))))
        rows.append(("chitu_amd ops.fp8_gemm_deepseek_v3", case, time_us(lambda: ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.bfloat16))))
    for case in hc.FUSED_MOE_FP8_CASES:
        x, w1, w2, w1s, w2s, ids, wts = [t.cuda() for t in hc.fused_moe_fp8_case(case)]
        rows.append(("chitu_amd fused_moe.fused_experts_impl fp8 block w8a8", case, time_us(
            lambda: fused_moe.fused_experts_impl(x.clone(), w1, w2, wts, ids, inplace=False, use_fp8_w8a8=True, w1_scale=w1s,
                                                 w2_scale=w2s, block_shape=[128, 128]))))
    for case in hc.FUSED_MOE_BF16_CASES:
        x, w1, w2, ids, wts = [t.cuda() for t in hc.fused_moe_bf16_case(case)]
        rows.append(("chitu_amd fused_moe.fused_experts_impl bf16", case, time_us(
            lambda: fused_moe.fused_experts_impl(x.clone(), w1, w2, wts, ids, inplace=False, use_fp8_w8a8=False))))
    be = HipAttnBackend(local_n_heads=16)
    for case in hc.MLA_DECODE_CASES:
        cache, qn, qp, table, lens, scale = hc.mla_decode_case(case)
        cd, qn, qp, table, lens = cache.cuda(), qn.cuda(), qp.cuda(), table.cuda(), lens.cuda()
        rows.append(("chitu_amd HipAttnBackend.mla_decode", case, time_us(lambda: be.mla_decode(qn, qp, cd, lens, table, scale))))
    print("# the same cases on chitu_amd's ops (tools/hw_cases_bench.py; same timing method, host launch path included)")
    for k, c, us in rows:
        print(json.dumps({"kernel": k, "case": c, "us_per_call": round(us, 2)}))


if __name__ == "__main__":
    main()
