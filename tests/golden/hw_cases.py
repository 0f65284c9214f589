"""Seeded inputs of the HARDWARE goldens (tests/golden/hw_*.npz): the same tensors are rebuilt here by the generator
(tests/golden/gen_hw_golden.py: the reference's own Triton kernels, compiled by Triton-ROCm and run on an MI355X) and by
the tests that compare against the fixtures -- the fixtures store outputs (and timings), not the multi-megabyte inputs.
CPU torch generators only: bit-identical on every machine.  Shapes are DeepSeek-R1 at TP=8 per rank unless noted."""
import numpy as np
import torch

FP8 = torch.float8_e4m3fn


def fp8_linear_case(name):
    """act_quant_deepseek_v3 + fp8_gemm_deepseek_v3 (chitu/ops.py:330-353, 453-483).  Returns x bf16, w fp8, ws f32."""
    M, N, K, seed = {"wqkv_a": (16, 2112, 7168, 101), "wo": (16, 7168, 2048, 102), "wq_b_m64": (64, 3072, 1536, 103),
                     "ragged": (5, 384, 512, 104)}[name]
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, K, generator=g, dtype=torch.float32) * 0.7).to(torch.bfloat16)
    if name == "ragged":
        x[3, 128:256] *= 40.0  # one hot group
    w = (torch.randn(N, K, generator=g, dtype=torch.float32) * 0.5).to(FP8)
    ws = torch.rand((N + 127) // 128, K // 128, generator=g, dtype=torch.float32) * 0.02 + 0.01
    return x, w, ws


FP8_LINEAR_CASES = ("wqkv_a", "wo", "wq_b_m64", "ragged")


def fused_moe_fp8_case(name):
    """fused_experts_impl(use_fp8_w8a8=True, block_shape=[128, 128]) (chitu/fused_moe.py:1130-1307): R1's per-rank expert
    shapes (W1 [512, 7168], W2 [7168, 256]) with 32 experts so that the inputs rebuild in seconds; "small" = the
    interpreter fixture's configuration (tests/golden/fused_moe_fp8.npz) with fresh values."""
    M, E, topk, K, I, seed = {"r1_bs16": (16, 32, 8, 7168, 256, 201), "r1_bs1": (1, 32, 8, 7168, 256, 202),
                              "small": (6, 8, 2, 256, 128, 203)}[name]
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, K, generator=g, dtype=torch.float32) * 0.5).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * I, K, generator=g, dtype=torch.float32) * 0.5).to(FP8)
    w2 = (torch.randn(E, K, I, generator=g, dtype=torch.float32) * 0.5).to(FP8)
    w1s = torch.rand(E, 2 * I // 128, K // 128, generator=g, dtype=torch.float32) * 0.02 + 0.01
    w2s = torch.rand(E, K // 128, I // 128, generator=g, dtype=torch.float32) * 0.02 + 0.01
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)])
    wts = torch.rand(M, topk, generator=g, dtype=torch.float32).to(torch.bfloat16)
    return x, w1, w2, w1s, w2s, ids, wts


FUSED_MOE_FP8_CASES = ("r1_bs16", "r1_bs1", "small")


def fused_moe_bf16_case(name):
    """fused_experts_impl(use_fp8_w8a8=False) on bf16 weights (the soft-fp8 / unquantised branch, model_deepseek_v3.py:975-993)."""
    M, E, topk, K, I, seed = {"bs16": (16, 16, 4, 2048, 256, 301), "small": (7, 8, 3, 256, 128, 302)}[name]
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn(M, K, generator=g, dtype=torch.float32) * 0.5).to(torch.bfloat16)
    w1 = (torch.randn(E, 2 * I, K, generator=g, dtype=torch.float32) * 0.1).to(torch.bfloat16)
    w2 = (torch.randn(E, K, I, generator=g, dtype=torch.float32) * 0.1).to(torch.bfloat16)
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(M)])
    wts = torch.rand(M, topk, generator=g, dtype=torch.float32).to(torch.bfloat16)
    return x, w1, w2, ids, wts


FUSED_MOE_BF16_CASES = ("bs16", "small")


def mla_decode_case(name):
    """mla_decode (chitu/triton_decode_attention.py:259-290; called from attn_backend.py:707-774 with 4 KV splits) on bf16
    inputs: 16 local heads, latent 512 + rope 64, 64-token pages, ragged contexts."""
    lens, seed = {"ragged": ([1, 77, 200, 1000], 401), "ctx4k": ([4096, 3000, 64, 65], 402)}[name]
    g = torch.Generator().manual_seed(seed)
    bs, H, C, R, page = len(lens), 16, 512, 64, 64
    per = [(n + page - 1) // page for n in lens]
    pages = sum(per) + 3
    perm = torch.randperm(pages, generator=g)
    table = torch.zeros(bs, max(per), dtype=torch.int32)
    k = 0
    for b, n in enumerate(per):
        table[b, :n] = perm[k : k + n].to(torch.int32)
        k += n
    cache = torch.randn(pages, page, C + R, generator=g, dtype=torch.float32).to(torch.bfloat16)
    q_nope = (torch.randn(bs, H, C, generator=g, dtype=torch.float32) * 0.3).to(torch.bfloat16)
    q_pe = (torch.randn(bs, H, R, generator=g, dtype=torch.float32) * 0.3).to(torch.bfloat16)
    return cache, q_nope, q_pe, table, torch.tensor(lens, dtype=torch.int32), 0.1352


MLA_DECODE_CASES = ("ragged", "ctx4k")


def gqa_decode_case(name):
    """RefAttnBackend.attn_with_kvcache (chitu/attn_backend.py:457-516) on contiguous caches [B, S, Hkv, 128]: Llama-3-8B's
    head counts (32 query / 8 KV heads), random bf16 values, ragged lengths around the 256-token page edge."""
    B, S, Hq, Hkv, lens, seed = {"llama3": (4, 1100, 32, 8, [0, 255, 256, 1024], 501), "mha": (2, 300, 8, 8, [7, 299], 502)}[name]
    g = torch.Generator().manual_seed(seed)
    D = 128
    kc = torch.randn(B, S, Hkv, D, generator=g, dtype=torch.float32).to(torch.bfloat16)
    vc = torch.randn(B, S, Hkv, D, generator=g, dtype=torch.float32).to(torch.bfloat16)
    q = (torch.randn(B, 1, Hq, D, generator=g, dtype=torch.float32) * 0.5).to(torch.bfloat16)
    kn = torch.randn(B, 1, Hkv, D, generator=g, dtype=torch.float32).to(torch.bfloat16)
    vn = torch.randn(B, 1, Hkv, D, generator=g, dtype=torch.float32).to(torch.bfloat16)
    return q, kc, vc, kn, vn, torch.tensor(lens, dtype=torch.int64)


GQA_DECODE_CASES = ("llama3", "mha")


def gqa_prefill_case(name):
    """RefAttnBackend.attn_varlen_func (chitu/attn_backend.py:394-455) as Attention.prefill_forward calls it
    (models/model.py:104-132): causal GQA over ragged prompts, head_dim 128."""
    seqs, Hq, Hkv, seed = {"llama3": ([1, 255, 257, 9, 600], 32, 8, 601), "g2": ([130, 64], 8, 4, 602)}[name]
    g = torch.Generator().manual_seed(seed)
    T = sum(seqs)
    cu = torch.tensor([0] + list(np.cumsum(seqs)), dtype=torch.int32)
    q = (torch.randn(T, Hq, 128, generator=g, dtype=torch.float32) * 0.5).to(torch.bfloat16)
    k = torch.randn(T, Hkv, 128, generator=g, dtype=torch.float32).to(torch.bfloat16)
    v = torch.randn(T, Hkv, 128, generator=g, dtype=torch.float32).to(torch.bfloat16)
    return q, k, v, cu, seqs


GQA_PREFILL_CASES = ("llama3", "g2")


def gqa_prefill_rows(seqs):
    """Token rows of a prefill case the fixture keeps (it stays small): the first / last / page- and tile-edge tokens of every
    sequence and a stride through the rest."""
    keep, s0 = set(), 0
    for n in seqs:
        keep.update(s0 + i for i in (0, 1, 31, 32, 63, 64, 127, 128, 254, 255, 256, n - 2, n - 1) if 0 <= i < n)
        s0 += n
    keep.update(range(0, s0, 23))
    return np.array(sorted(keep), dtype=np.int64)


def soft_fp8_moe_case(name):
    """The non-NVIDIA soft-fp8 MoE branch (chitu/models/model_deepseek_v3.py:975-996): weight_dequant_soft_fp8_deepseek_v3 of the
    stacked fp8 experts (chitu/ops.py:396-449), then fused_experts(use_fp8_w8a8=False) on the bf16 result.  Inputs: the fp8
    case of the same name's weights and scales."""
    return fused_moe_fp8_case({"r1_bs16": "r1_bs16", "small": "small"}[name])


SOFT_FP8_MOE_CASES = ("r1_bs16", "small")


def bits16(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits8(t):
    return t.detach().cpu().contiguous().view(torch.uint8).numpy()


def from_bits16(a):
    return torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16)
