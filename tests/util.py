"""Shared test helpers (fixture loading, bit views)."""

import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def bf16(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16)


def fp8(bits):
    return torch.from_numpy(np.ascontiguousarray(bits)).view(torch.float8_e4m3fn)


def bits16(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits8(t):
    return t.detach().cpu().contiguous().view(torch.uint8).numpy()


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def max_rel_to_peak(a, b):
    """max |a-b| / max|b|: the '<=1e-2 rel' bar of BASELINE.md for fp8 GEMM / attention."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def assert_close_elementwise(a, b, rtol=1e-2, atol_peak=2e-3, what="", outlier_frac=0.0):
    """Element-wise bar beside max_rel_to_peak: |a - b| <= rtol * |b| + atol_peak * max|b| for EVERY element -- the
    north_star's "1e-2 rel" read per element, with an absolute floor (a fraction of the tensor's peak) for the elements
    near zero, whose error is set by their neighbours' magnitude (fp32 summation order, one bf16 rounding), not by
    their own.  The reference's own test uses allclose(rtol=atol=5e-3) on O(1) data (test/pytest/test_w8a8.py:29)."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    bound = rtol * b.abs() + atol_peak * b.abs().max()
    bad = (a - b).abs() > bound
    # outlier_frac: the share of elements allowed outside this bar (they stay under the caller's PEAK bar) -- for paths
    # with an fp8 re-quantisation in the middle, where one flipped code of an intermediate moves a few outputs by more
    # than 1 % of their own value while staying well inside 1 % of the peak
    if int(bad.sum()) > outlier_frac * bad.numel():
        i = int(((a - b).abs() - bound).argmax())
        raise AssertionError(f"{what}: {int(bad.sum())} of {bad.numel()} elements outside rtol={rtol} + {atol_peak}*peak; "
                             f"worst: got {a.flatten()[i].item()}, want {b.flatten()[i].item()}, peak {b.abs().max().item()}")


def assert_close(a, b, peak_tol, rtol=1e-2, atol_frac=None, what="", outlier_frac=0.0):
    """The two bars together: max|a - b| < peak_tol * max|b| (BASELINE's "1e-2 rel" in the peak norm) AND, for every
    element, |a - b| <= rtol * |b| + atol_frac * peak_tol * max|b| -- 1 % of the element's own value plus half of the
    peak bar as the absolute floor wherever the bar is the north_star's (peak_tol <= 1e-2: op-level comparisons against
    the oracle).  Looser bars are model-level comparisons (a tiny random model amplifies one flipped bf16 per layer) or
    fixtures carrying the Triton interpreter's cast defects: there the element-wise form adds nothing to the peak bar.
    Returns the peak-normalised error."""
    if atol_frac is None:
        atol_frac = 0.5 if peak_tol <= 1e-2 else 1.0
    err = max_rel_to_peak(a, b)
    assert err < peak_tol, (what, err)
    assert_close_elementwise(a, b, rtol=rtol, atol_peak=peak_tol * atol_frac, what=what, outlier_frac=outlier_frac)
    return err


def pattern_cache(pages, page, dim):
    p = torch.arange(pages).view(-1, 1, 1)
    s = torch.arange(page).view(1, -1, 1)
    d = torch.arange(dim).view(1, 1, -1)
    return (((p * 64 + s) % 251).float() * 0.25 + (d % 7).float() - 3.0).to(torch.bfloat16)


def lattice(*shape, mod=97, scale=32.0, salt=0):
    """Same deterministic bf16-exact values as tests/golden/gen_golden.py::lattice (fixture inputs are
    recomputed, not stored)."""
    primes = [131, 7, 53, 3, 17]
    acc = torch.zeros(shape, dtype=torch.int64) + salt
    for ax, n in enumerate(shape):
        view = [1] * len(shape)
        view[ax] = n
        acc = acc + torch.arange(n).view(view) * primes[ax]
    return ((acc % mod - mod // 2).float() / scale).to(torch.bfloat16)


def gqa_golden_case():
    """Inputs of tests/golden/gqa_decode.npz laid out in shuffled pages of 256 (+ the contiguous form)."""
    g = golden("gqa_decode")
    B, S, Hq, Hkv, D = [int(v) for v in g["dims"]]
    lens = torch.from_numpy(g["lens"]).to(torch.int32)
    kc, vc = lattice(B, S, Hkv, D, salt=1), lattice(B, S, Hkv, D, salt=5)
    q = lattice(B, 1, Hq, D, mod=89, scale=64.0, salt=11)
    k_new, v_new = lattice(B, 1, Hkv, D, mod=83, salt=3), lattice(B, 1, Hkv, D, mod=79, salt=9)
    page = 256
    per = (S + page - 1) // page
    perm = torch.randperm(B * per + 3, generator=torch.Generator().manual_seed(4))
    table = perm[: B * per].view(B, per).to(torch.int32)
    k_pages = torch.zeros(B * per + 3, page, Hkv, D, dtype=torch.bfloat16)
    v_pages = torch.zeros_like(k_pages)
    for b in range(B):
        for p in range(per):
            n = min(page, S - p * page)
            k_pages[table[b, p], :n] = kc[b, p * page : p * page + n]
            v_pages[table[b, p], :n] = vc[b, p * page : p * page + n]
    return dict(q=q, k_new=k_new, v_new=v_new, lens=lens, table=table, k_pages=k_pages, v_pages=v_pages,
                out=bf16(g["out"]), D=D)


def mla_prefill_golden_case():
    g = golden("mla_prefill")
    H, C, R = [int(v) for v in g["dims"]]
    seqs = [int(v) for v in g["seqs"]]
    T = sum(seqs)
    cu = torch.tensor([0] + list(np.cumsum(seqs)), dtype=torch.int32)
    kv = lattice(T, 1, C + R, salt=2)
    q = lattice(T, H, C + R, mod=89, scale=128.0, salt=13)
    return dict(q=q, kv=kv, cu=cu, seqs=seqs, scale=float(g["scale"][0]), rows=torch.from_numpy(g["rows"]), out=bf16(g["out"]), C=C)


def gqa_prefill_golden_case():
    g = golden("gqa_prefill")
    seqs = [int(v) for v in g["seqs"]]
    T = sum(seqs)
    cu = torch.tensor([0] + list(np.cumsum(seqs)), dtype=torch.int32)
    return dict(q=lattice(T, 8, 128, mod=89, scale=64.0, salt=4), k=lattice(T, 2, 128, salt=6), v=lattice(T, 2, 128, mod=83, salt=8),
                cu=cu, seqs=seqs, rows=torch.from_numpy(g["rows"]), out=bf16(g["out"]))


def ref_model_case():
    """Parameters of tests/golden/ref_model_v3.npz regenerated exactly as tests/golden/gen_ref_model.py::fill
    does (same seed, same sorted-name order), keyed by the oracle's / chitu_amd's parameter names."""
    g = golden("ref_model_v3")
    # the reference model's constructor leaves torch's default dtype at bfloat16 (model_deepseek_v3.py:1131), so
    # gen_ref_model.py::fill drew its randn / rand tensors in bf16: same here
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        return g, _ref_model_params(g)
    finally:
        torch.set_default_dtype(prev)


def _ref_model_params(g):
    gen = torch.Generator().manual_seed(1234)
    dts = {"torch.float8_e4m3fn": torch.float8_e4m3fn, "torch.float32": torch.float32, "torch.bfloat16": torch.bfloat16}
    params = {}
    for n, shp, dt in zip(g["names"].tolist(), g["shapes"].tolist(), g["dtypes"].tolist()):
        shape, dtype = eval(shp), dts[dt]
        if dtype == torch.float8_e4m3fn:
            t = (torch.randn(shape, generator=gen) * 0.5).to(torch.float8_e4m3fn)
        elif dtype == torch.float32:
            t = torch.rand(shape, generator=gen) * 0.02 + 0.01
        elif n.endswith("norm.weight"):
            t = torch.ones(shape, dtype=dtype)
        elif n.endswith("gate.weight"):
            t = (torch.randn(shape, generator=gen) * shape[-1] ** -0.5).to(dtype)
        else:
            t = (torch.randn(shape, generator=gen) * 0.05).to(dtype)
        # the reference stacks the experts as module `w1w3` / `w2` with .weight/.scale; here they are flat names
        if t.dim() == 3:  # MoE layers only; the dense MLP keeps w1w3.weight / w2.weight
            for m in ("w1w3", "w2"):
                n = n.replace(f"ffn.{m}.weight", f"ffn.{m}_weight").replace(f"ffn.{m}.scale", f"ffn.{m}_scale")
        params[n] = t.to(dtype) if t.dtype != dtype else t
    return params


CKPT_TINY = dict(name="tiny-ckpt", type="deepseek-v3", vocab_size=512, dim=256, inter_dim=512, moe_inter_dim=256, n_layers=2,
                 n_dense_layers=1, n_heads=4, n_routed_experts=4, n_shared_experts=1, n_activated_experts=2,
                 n_expert_groups=1, n_limited_groups=1, route_scale=2.5, score_func="sigmoid", q_lora_rank=128,
                 kv_lora_rank=128, qk_nope_head_dim=128, qk_rope_head_dim=64, v_head_dim=128, rope_theta=10000.0,
                 rope_factor=40, main_weight_dtype="float8_e4m3fn")


def tiny_hf_checkpoint(cfg=CKPT_TINY, seed=77):
    """A DeepSeek-V3 checkpoint under HUGGING FACE names (FP8 weights + `weight_scale_inv` block scales, bf16 norms /
    router / embeddings, the router's e_score_correction_bias, and one tensor of the multi-token-prediction layer
    `model.layers.61` that the loader must drop).  Shared by tests/golden/gen_ckpt.py (fed to the reference's loader)
    and the checkpoint tests (fed to chitu_amd.checkpoint)."""
    g = torch.Generator().manual_seed(seed)
    c = cfg
    H, qk = c["n_heads"], c["qk_nope_head_dim"] + c["qk_rope_head_dim"]
    sd = {}

    def fp8w(name, n, k):
        sd[name + ".weight"] = (torch.randn(n, k, generator=g, dtype=torch.float32) * 0.5).to(torch.float8_e4m3fn)
        sd[name + ".weight_scale_inv"] = torch.rand((n + 127) // 128, (k + 127) // 128, generator=g, dtype=torch.float32) * 0.02 + 0.01

    def bf(name, *shape):
        sd[name] = (torch.randn(*shape, generator=g, dtype=torch.float32) * 0.05).to(torch.bfloat16)

    def mlp(prefix, inter):
        fp8w(prefix + "gate_proj", inter, c["dim"])
        fp8w(prefix + "up_proj", inter, c["dim"])
        fp8w(prefix + "down_proj", c["dim"], inter)

    bf("model.embed_tokens.weight", c["vocab_size"], c["dim"])
    for i in range(c["n_layers"]):
        p = f"model.layers.{i}."
        bf(p + "input_layernorm.weight", c["dim"])
        bf(p + "post_attention_layernorm.weight", c["dim"])
        a = p + "self_attn."
        fp8w(a + "q_a_proj", c["q_lora_rank"], c["dim"])
        bf(a + "q_a_layernorm.weight", c["q_lora_rank"])
        fp8w(a + "q_b_proj", H * qk, c["q_lora_rank"])
        fp8w(a + "kv_a_proj_with_mqa", c["kv_lora_rank"] + c["qk_rope_head_dim"], c["dim"])
        bf(a + "kv_a_layernorm.weight", c["kv_lora_rank"])
        fp8w(a + "kv_b_proj", H * (c["qk_nope_head_dim"] + c["v_head_dim"]), c["kv_lora_rank"])
        fp8w(a + "o_proj", c["dim"], H * c["v_head_dim"])
        if i < c["n_dense_layers"]:
            mlp(p + "mlp.", c["inter_dim"])
        else:
            bf(p + "mlp.gate.weight", c["n_routed_experts"], c["dim"])
            bf(p + "mlp.gate.e_score_correction_bias", c["n_routed_experts"])
            for e in range(c["n_routed_experts"]):
                mlp(p + f"mlp.experts.{e}.", c["moe_inter_dim"])
            mlp(p + "mlp.shared_experts.", c["moe_inter_dim"] * c["n_shared_experts"])
    bf("model.norm.weight", c["dim"])
    bf("lm_head.weight", c["vocab_size"], c["dim"])
    bf("model.layers.61.embed_tokens.weight", 8, c["dim"])
    return sd


HF_LLAMA_TINY = dict(dim=64, n_layers=2, n_heads=4, n_kv_heads=2, vocab_size=32, ffn_dim=96, num_local_experts=4)


def tiny_hf_llama_checkpoint(kind="llama", cfg=HF_LLAMA_TINY, seed=91):
    """A Llama-family checkpoint under HUGGING FACE names (bf16), for the loaders of configs 1 / 2 / 4:
      "llama":   separate q / k / v and gate / up projections (GQA), no biases -- Llama-2 / Llama-3;
      "merged":  qkv_proj / gate_up_proj already merged, with a qkv bias -- the Phi-3 / GLM-style files the reference
                 splits before tensor-parallel sharding (model_hf_llama.py:428-504);
      "mixtral": block_sparse_moe.gate + experts.{e}.w1 / w3 / w2 (model_hf_mixtral.py:171-178).
    Shared by tests/golden/gen_ckpt_llama.py (fed to the reference's loader methods) and tests/test_checkpoint.py."""
    g = torch.Generator().manual_seed(seed + {"llama": 0, "merged": 1, "mixtral": 2}[kind])
    c = cfg
    hd = c["dim"] // c["n_heads"]
    sd = {}

    def bf(name, *shape):
        sd[name] = (torch.randn(*shape, generator=g, dtype=torch.float32) * 0.05).to(torch.bfloat16)

    bf("model.embed_tokens.weight", c["vocab_size"], c["dim"])
    for i in range(c["n_layers"]):
        p = f"model.layers.{i}."
        bf(p + "input_layernorm.weight", c["dim"])
        bf(p + "post_attention_layernorm.weight", c["dim"])
        a = p + "self_attn."
        if kind == "merged":
            bf(a + "qkv_proj.weight", (c["n_heads"] + 2 * c["n_kv_heads"]) * hd, c["dim"])
            bf(a + "qkv_proj.bias", (c["n_heads"] + 2 * c["n_kv_heads"]) * hd)
        else:
            bf(a + "q_proj.weight", c["n_heads"] * hd, c["dim"])
            bf(a + "k_proj.weight", c["n_kv_heads"] * hd, c["dim"])
            bf(a + "v_proj.weight", c["n_kv_heads"] * hd, c["dim"])
        bf(a + "o_proj.weight", c["dim"], c["n_heads"] * hd)
        if kind == "mixtral":
            m = p + "block_sparse_moe."
            bf(m + "gate.weight", c["num_local_experts"], c["dim"])
            for e in range(c["num_local_experts"]):
                bf(m + f"experts.{e}.w1.weight", c["ffn_dim"], c["dim"])
                bf(m + f"experts.{e}.w3.weight", c["ffn_dim"], c["dim"])
                bf(m + f"experts.{e}.w2.weight", c["dim"], c["ffn_dim"])
        elif kind == "merged":
            bf(p + "mlp.gate_up_proj.weight", 2 * c["ffn_dim"], c["dim"])
            bf(p + "mlp.down_proj.weight", c["dim"], c["ffn_dim"])
        else:
            bf(p + "mlp.gate_proj.weight", c["ffn_dim"], c["dim"])
            bf(p + "mlp.up_proj.weight", c["ffn_dim"], c["dim"])
            bf(p + "mlp.down_proj.weight", c["dim"], c["ffn_dim"])
    bf("model.norm.weight", c["dim"])
    bf("lm_head.weight", c["vocab_size"], c["dim"])
    return sd


def tensor_digest(t):
    import hashlib

    raw = t.detach().contiguous().view(torch.uint8).numpy().tobytes()
    return [list(t.shape), str(t.dtype), hashlib.sha1(raw).hexdigest()]


# ---------------------------------------------------------------- BASELINE config 1: the reference's Llama on CPU
def ref_llama_fixture():
    """tests/golden/ref_llama.npz (tests/golden/gen_ref_llama.py: the reference's TransformerLlama, bs 1, prefill +
    64 greedy decode steps on CPU) plus the parameters, regenerated from the generator's seed in its sorted-name
    order and mapped onto chitu_amd.llama / oracle.llama names (wq|wk|wv -> wqkv, w1|w3 -> w13) by the product loader
    chitu_amd.checkpoint.preprocess_meta_llama -- so the reference-vs-HIP Llama tests also cover that loader."""
    import ast

    g = golden("ref_llama")
    cfg = ast.literal_eval(str(g["cfg"][0]))
    gen = torch.Generator().manual_seed(4321)  # gen_ref_llama.py::SEED / fill
    ref = {}
    for name, shape in zip(g["names"].tolist(), g["shapes"].tolist()):
        shape = ast.literal_eval(shape)
        if name.endswith("norm.weight"):
            t = (1.0 + 0.1 * torch.randn(shape, generator=gen, dtype=torch.float32)).to(torch.bfloat16)
        elif name.startswith("tok_embeddings"):
            t = torch.randn(shape, generator=gen, dtype=torch.float32).to(torch.bfloat16)
        else:
            t = (torch.randn(shape, generator=gen, dtype=torch.float32) * shape[-1] ** -0.5).to(torch.bfloat16)
        ref[name] = t
    from chitu_amd.checkpoint import preprocess_meta_llama

    return g, cfg, preprocess_meta_llama(ref)  # the product loader: Meta names -> LlamaDecoder parameters


# ---------------------------------------------------------------- cache manager scenario
def cache_manager_scenario(make_manager, make_varlens):
    """One scripted life of a paged cache manager (prefill of ragged prompts, decode steps across page boundaries with
    the kernels' in-place append simulated through the manager's own block table, a finished request, its pages reused
    by a new one), observed only through quantities that do not depend on WHICH physical pages were handed out:
    sequence lengths, the device length buffers, pages per request, free-page count, distinctness of live pages, the
    device block table == the host one, and a digest of every request's logical cache content per layer.
    Run on the reference's PagedKVCacheManager by tests/golden/gen_cache_manager.py and on chitu_amd's by the test."""
    import hashlib

    layers, page, width = 2, 4, 8
    mgr = make_manager(layers=layers, page=page, width=width, max_reqs=3, max_seq_len=16)
    obs = []
    counter = [0]

    def rows(n):  # deterministic, distinct payload rows
        base = counter[0]
        counter[0] += n
        return (torch.arange(base * width, (base + n) * width, dtype=torch.float32).view(n, width) % 251).to(torch.bfloat16)

    def gathered(req, layer):
        ids, n = mgr.block_table[req], mgr.seq_lens[req]
        cache = mgr.get_paged_kv_cache(layer)
        parts = [cache[b][: min(page, n - i * page)] for i, b in enumerate(ids) if n - i * page > 0]
        flat = torch.cat(parts).contiguous() if parts else torch.zeros(0, width, dtype=torch.bfloat16)
        return hashlib.sha1(flat.view(torch.int16).numpy().tobytes()).hexdigest()

    def observe(tag, live):
        live_pages = [b for r in live for b in mgr.block_table[r]]
        obs.append({
            "tag": tag,
            "seq_lens": sorted((r, int(mgr.seq_lens[r])) for r in live),
            "pages": sorted((r, len(mgr.block_table[r])) for r in live),
            "free": len(mgr.free_blocks),
            "pages_distinct": len(set(live_pages)) == len(live_pages),
            "content": sorted((r, l, gathered(r, l)) for r in live for l in range(layers)),
        })

    def prefill(reqs, lens):
        vl = make_varlens([[0] * n for n in lens])
        mgr.curr_varlens, mgr.curr_req_ids = vl, reqs
        for layer in range(layers):
            data = rows(sum(lens))
            mgr.finalize_cache_bylayer_prefill(data, None, reqs, vl, layer)
        mgr.finalize_cache_all_prefill(reqs, vl)

    def decode(reqs, tag):
        mgr.prepare_cache_decode(reqs)
        mgr.prepare_block_table_for_decode(reqs)
        n = len(reqs)
        excl = mgr.get_gpu_seq_lens_excl_this_decode().tolist()
        incl = mgr.get_gpu_seq_lens_incl_this_decode().tolist()
        table = mgr.get_gpu_block_table()
        same_table = all(table[i, : len(mgr.block_table[r])].tolist() == list(mgr.block_table[r]) for i, r in enumerate(reqs))
        for layer in range(layers):  # what append_to_paged_kv_cache / the attention kernels do in place
            new = rows(n)
            cache = mgr.get_paged_kv_cache(layer)
            for i, r in enumerate(reqs):
                L = excl[i]
                cache[int(table[i, L // page])][L % page] = new[i]
        mgr.finalize_cache_single_decode(reqs)
        observe(tag, reqs)
        obs[-1].update(excl=excl, incl=incl, table_rows=int(table.shape[0]), device_table_matches=same_table)

    prefill(["a", "b"], [5, 2])
    observe("prefill ab", ["a", "b"])
    for step in range(4):  # a: 5 -> 9 (new page at 8), b: 2 -> 6 (new page at 4)
        decode(["a", "b"], f"decode ab {step}")
    mgr.finalize_cache_all_decode("b")
    observe("b finished", ["a"])
    prefill(["c"], [3])
    observe("prefill c", ["a", "c"])
    for step in range(2):
        decode(["c", "a"], f"decode ca {step}")
    return obs


# ---------------------------------------------------------------- reads of memory nobody wrote
import contextlib  # noqa: E402


@contextlib.contextmanager
def poisoned_allocations():
    """torch.empty / empty_like / Tensor.new_empty on the GPU return memory filled with 0xFF bytes (NaN as bf16 / fp32 /
    e4m3fn, -1 as int32) and every workspace.get() re-fills its buffer: a launch that READS what no launch of the step
    wrote reads NaN / -1, and the step's output changes.  (An `empty` buffer normally holds whatever its previous user
    left -- usually the same step's own values of the previous iteration, which hides such a read.)"""
    from chitu_amd import workspace

    real_empty, real_like, real_new, real_ws = torch.empty, torch.empty_like, torch.Tensor.new_empty, workspace.get

    def foul(t):
        if isinstance(t, torch.Tensor) and t.is_cuda and t.numel():
            if t.is_contiguous():
                t.view(-1).view(torch.uint8).fill_(0xFF)
            else:
                t.fill_(float("nan") if t.is_floating_point() else -1)
        return t

    def empty(*a, **k):
        return foul(real_empty(*a, **k))

    def empty_like(*a, **k):
        return foul(real_like(*a, **k))

    def new_empty(self, *a, **k):
        return foul(real_new(self, *a, **k))

    def ws_get(nbytes, device, tag="default"):
        return foul(real_ws(nbytes, device, tag))

    torch.empty, torch.empty_like, torch.Tensor.new_empty, workspace.get = empty, empty_like, new_empty, ws_get
    try:
        yield
    finally:
        torch.empty, torch.empty_like, torch.Tensor.new_empty, workspace.get = real_empty, real_like, real_new, real_ws
