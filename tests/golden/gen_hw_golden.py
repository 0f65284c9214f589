"""Goldens from the REFERENCE'S OWN Triton kernels executed ON THE MI355X (real Triton-ROCm: RTNE fp8 / bf16 casts, bf16
tl.dot -- none of the Triton interpreter's cast defects that the CPU fixtures of gen_golden.py carry), with their timings.

    CHITU_REFERENCE_DIR=<checkout of thu-pacman/chitu> python tests/golden/gen_hw_golden.py [name ...]

Runs on a GPU box that has a copy of the reference tree (the build stages one beside the repo snapshot for this call; it is
never committed).  Writes tests/golden/hw_<name>.npz (outputs only: the inputs are rebuilt from seeds by
tests/golden/hw_cases.py) and prints / writes the timing table (profiles/r05_reference_triton_on_mi355x.txt when
HW_GOLDEN_PROFILE is set).  Kernels that do not compile for gfx950 are listed in the table with the error, not worked around.
Reference entry points exercised (all unmodified):
  chitu/ops.py:330-353            act_quant_deepseek_v3          (triton_kernels.py:193-214)
  chitu/ops.py:453-483            fp8_gemm_deepseek_v3           (triton_kernels.py:303-388, autotuned)
  chitu/fused_moe.py:1130-1307    fused_experts_impl             (fused_moe_kernel :62-307, moe_align stages :314-442,
                                                                   per_token_group_quant_fp8 :640-720, SiluAndMul :24-39)
  chitu/triton_decode_attention.py:259-290  mla_decode           (_mla_attn_kernel :21-130, _mla_softmax_reducev_kernel :185-232)
"""
import json
import os
import sys
import time
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import hw_cases as hc  # noqa: E402

TIMES = []


def install_reference():
    ref = os.environ.get("CHITU_REFERENCE_DIR", "")
    assert os.path.isdir(os.path.join(ref, "chitu")), "CHITU_REFERENCE_DIR must point at a checkout of the reference"
    os.environ.pop("TRITON_INTERPRET", None)
    sys.dont_write_bytecode = True
    sys.path.insert(0, ref)
    for name in ("chitu_backend", "tiktoken", "tiktoken.load"):
        m = types.ModuleType(name)
        if name == "tiktoken.load":
            m.load_tiktoken_bpe = lambda *a, **k: {}
        sys.modules.setdefault(name, m)
    import chitu.device_type as dt

    dt._device_name = "AMD Instinct MI355X"  # neither NVIDIA nor muxi: moe_align takes the Triton stages (fused_moe.py:605)


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


def record(kernel, case, shape, us=None, error=None):
    row = {"kernel": kernel, "case": case, "shape": shape}
    if us is not None:
        row["us_per_call"] = round(us, 2)
    if error is not None:
        row["error"] = error[:300]
    TIMES.append(row)
    print(json.dumps(row), flush=True)


def save(name, **arrs):
    path = os.path.join(HERE, "hw_" + name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, {k: getattr(v, "shape", None) for k, v in arrs.items()}, flush=True)


def gen_fp8_linear():
    from chitu import ops

    out = {}
    torch.set_default_dtype(torch.bfloat16)
    try:
        for case in hc.FP8_LINEAR_CASES:
            x, w, ws = hc.fp8_linear_case(case)
            xd, wd, wsd = x.cuda(), w.cuda(), ws.cuda()
            shape = f"M={x.shape[0]} N={w.shape[0]} K={w.shape[1]}"
            try:
                xq, xs = ops.act_quant_deepseek_v3(xd, 128)
                record("act_quant_deepseek_v3", case, shape, time_us(lambda: ops.act_quant_deepseek_v3(xd, 128)))
                c = ops.fp8_gemm_deepseek_v3(xq, xs, wd, wsd)
                record("fp8_gemm_deepseek_v3 (autotuned)", case, shape, time_us(lambda: ops.fp8_gemm_deepseek_v3(xq, xs, wd, wsd)))
                out[f"{case}_xq"], out[f"{case}_xs"], out[f"{case}_c"] = hc.bits8(xq), xs.float().cpu().numpy(), hc.bits16(c)
                if case == "ragged":
                    out["ragged_w_dequant"] = hc.bits16(ops.weight_dequant_deepseek_v3(wd, wsd, 128))
            except Exception as exc:  # noqa: BLE001
                record("act_quant / fp8_gemm", case, shape, error=f"{type(exc).__name__}: {exc}")
    finally:
        torch.set_default_dtype(torch.float32)
    if out:
        save("fp8_linear", **out)


def gen_fused_moe_fp8():
    from chitu.fused_moe import fused_experts_impl

    out = {}
    for case in hc.FUSED_MOE_FP8_CASES:
        x, w1, w2, w1s, w2s, ids, wts = hc.fused_moe_fp8_case(case)
        d = [t.cuda() for t in (x, w1, w2, w1s, w2s, ids, wts)]
        shape = f"M={x.shape[0]} E={w1.shape[0]} topk={ids.shape[1]} K={x.shape[1]} I={w2.shape[2]}"
        run = lambda: fused_experts_impl(d[0].clone(), d[1], d[2], d[6], d[5], inplace=False, use_fp8_w8a8=True,  # noqa: E731
                                         w1_scale=d[3], w2_scale=d[4], block_shape=[128, 128])
        try:
            o = run()
            record("fused_experts_impl fp8 block w8a8", case, shape, time_us(run))
            out[f"{case}_out"] = hc.bits16(o)
        except Exception as exc:  # noqa: BLE001
            record("fused_experts_impl fp8 block w8a8", case, shape, error=f"{type(exc).__name__}: {exc}")
    if out:
        save("fused_moe_fp8", **out)


def gen_fused_moe_bf16():
    from chitu.fused_moe import fused_experts_impl

    out = {}
    for case in hc.FUSED_MOE_BF16_CASES:
        x, w1, w2, ids, wts = hc.fused_moe_bf16_case(case)
        d = [t.cuda() for t in (x, w1, w2, ids, wts)]
        shape = f"M={x.shape[0]} E={w1.shape[0]} topk={ids.shape[1]} K={x.shape[1]} I={w2.shape[2]}"
        run = lambda: fused_experts_impl(d[0].clone(), d[1], d[2], d[4], d[3], inplace=False, use_fp8_w8a8=False)  # noqa: E731
        try:
            o = run()
            record("fused_experts_impl bf16", case, shape, time_us(run))
            out[f"{case}_out"] = hc.bits16(o)
        except Exception as exc:  # noqa: BLE001
            record("fused_experts_impl bf16", case, shape, error=f"{type(exc).__name__}: {exc}")
    if out:
        save("fused_moe_bf16", **out)


def gen_mla_decode():
    from chitu.triton_decode_attention import mla_decode

    out = {}
    for case in hc.MLA_DECODE_CASES:
        cache, q_nope, q_pe, table, lens, scale = hc.mla_decode_case(case)
        cd, qn, qp, tb, ln = cache.cuda(), q_nope.cuda(), q_pe.cuda(), table.cuda(), lens.cuda()
        bs, H, C = q_nope.shape
        splits = 4  # attn_backend.py:729
        o = torch.zeros(bs, H, C, dtype=torch.bfloat16, device="cuda")
        logits = torch.empty(bs, H, splits, C + 1, dtype=torch.float32, device="cuda")
        shape = f"bs={bs} H={H} lens={lens.tolist()}"
        run = lambda: mla_decode(qn, qp, cd[..., :C], cd[..., C:], o, tb, ln, logits, splits, scale, cache.shape[1])  # noqa: E731
        try:
            run()
            torch.cuda.synchronize()
            out[f"{case}_out"] = hc.bits16(o)
            record("mla_decode (_mla_attn_kernel + _mla_softmax_reducev_kernel)", case, shape, time_us(run))
        except Exception as exc:  # noqa: BLE001
            record("mla_decode", case, shape, error=f"{type(exc).__name__}: {exc}")
    if out:
        save("mla_decode", **out)


def gen_gqa():
    """Round 6: the reference's pure-torch attention (RefAttnBackend) run ON THE MI355X -- torch's own bf16 / fp32 kernels on
    ROCm instead of the CPU's (tests/golden/gqa_decode.npz, gqa_prefill.npz are the CPU runs on lattice values)."""
    from chitu.attn_backend import RefAttnBackend

    be = RefAttnBackend()
    out = {}
    for case in hc.GQA_DECODE_CASES:
        q, kc, vc, kn, vn, lens = hc.gqa_decode_case(case)
        d = [t.cuda() for t in (q, kc, vc, kn, vn, lens)]
        shape = f"B={q.shape[0]} Hq={q.shape[2]} Hkv={kc.shape[2]} lens={lens.tolist()}"
        try:
            o = be.attn_with_kvcache(d[0], d[1].clone(), d[2].clone(), d[3], d[4], cache_seqlens=d[5], softmax_scale=128 ** -0.5)
            record("RefAttnBackend.attn_with_kvcache", case, shape,
                   time_us(lambda: be.attn_with_kvcache(d[0], d[1], d[2], d[3], d[4], cache_seqlens=d[5], softmax_scale=128 ** -0.5), n=5))
            out[f"decode_{case}_out"] = hc.bits16(o)
        except Exception as exc:  # noqa: BLE001
            record("RefAttnBackend.attn_with_kvcache", case, shape, error=f"{type(exc).__name__}: {exc}")
    for case in hc.GQA_PREFILL_CASES:
        q, k, v, cu, seqs = hc.gqa_prefill_case(case)
        d = [t.cuda() for t in (q, k, v, cu)]
        shape = f"seqs={seqs} Hq={q.shape[1]} Hkv={k.shape[1]}"
        try:
            run = lambda: be.attn_varlen_func(d[0], d[1], d[2], d[3], d[3], max(seqs), max(seqs), causal=True)  # noqa: E731
            o = run()
            record("RefAttnBackend.attn_varlen_func", case, shape, time_us(run, n=5))
            out[f"prefill_{case}_out"] = hc.bits16(o[torch.from_numpy(hc.gqa_prefill_rows(seqs)).cuda()])  # kept rows only
        except Exception as exc:  # noqa: BLE001
            record("RefAttnBackend.attn_varlen_func", case, shape, error=f"{type(exc).__name__}: {exc}")
    if out:
        save("gqa", **out)


def gen_soft_fp8_moe():
    """Round 6: the README configuration's MoE branch off NVIDIA (infer.soft_fp8=True): the reference's soft-fp8 dequant kernel
    over the stacked experts, then its bf16 fused_experts -- both compiled by Triton-ROCm, on the MI355X."""
    from chitu import ops
    from chitu.fused_moe import fused_experts_impl

    out = {}
    torch.set_default_dtype(torch.bfloat16)  # the dequantised tensor takes torch's default dtype (ops.py:433), bf16 under the Backend
    for case in hc.SOFT_FP8_MOE_CASES:
        x, w1, w2, w1s, w2s, ids, wts = hc.soft_fp8_moe_case(case)
        d = [t.cuda() for t in (x, w1, w2, w1s, w2s, ids, wts)]
        shape = f"M={x.shape[0]} E={w1.shape[0]} topk={ids.shape[1]} K={x.shape[1]} I={w2.shape[2]}"
        try:
            w1d = ops.weight_dequant_soft_fp8_deepseek_v3(d[1], d[3], 128)
            w2d = ops.weight_dequant_soft_fp8_deepseek_v3(d[2], d[4], 128)
            record("weight_dequant_soft_fp8_deepseek_v3 (stacked w1)", case, shape,
                   time_us(lambda: ops.weight_dequant_soft_fp8_deepseek_v3(d[1], d[3], 128), n=5))
            run = lambda: fused_experts_impl(d[0].clone(), w1d, w2d, d[6], d[5], inplace=False, use_fp8_w8a8=False)  # noqa: E731
            o = run()
            record("fused_experts_impl bf16 on soft-dequantised experts", case, shape, time_us(run))
            out[f"{case}_out"] = hc.bits16(o)
            if case == "small":
                out["small_w1_dequant"] = hc.bits16(w1d)
        except Exception as exc:  # noqa: BLE001
            record("soft-fp8 MoE branch", case, shape, error=f"{type(exc).__name__}: {exc}")
    torch.set_default_dtype(torch.float32)
    if out:
        save("soft_fp8_moe", **out)


GENS = {"gqa": gen_gqa, "soft_fp8_moe": gen_soft_fp8_moe, "fp8_linear": gen_fp8_linear, "fused_moe_fp8": gen_fused_moe_fp8, "fused_moe_bf16": gen_fused_moe_bf16,
        "mla_decode": gen_mla_decode}

if __name__ == "__main__":
    install_reference()
    import triton

    names = sys.argv[1:] or list(GENS)
    t0 = time.time()
    for n in names:
        GENS[n]()
    meta = {"device": torch.cuda.get_device_name(0), "triton": triton.__version__, "torch": torch.__version__,
            "TRITON_INTERPRET": os.environ.get("TRITON_INTERPRET"), "seconds": round(time.time() - t0, 1)}
    print(json.dumps(meta), flush=True)
    prof = os.environ.get("HW_GOLDEN_PROFILE")
    if prof:
        with open(prof, "a" if os.environ.get("HW_GOLDEN_PROFILE_APPEND") else "w") as f:
            f.write("# The reference's own Triton kernels (thu-pacman/chitu, unmodified) compiled by Triton-ROCm and run on the MI355X:\n")
            f.write("# tests/golden/gen_hw_golden.py -- the run that produced tests/golden/hw_*.npz.  us per call, HIP events over 20 calls.\n")
            f.write("# " + json.dumps(meta) + "\n")
            for row in TIMES:
                f.write(json.dumps(row) + "\n")
