"""The REFERENCE'S OWN model classes running on chitu_amd's operator surface, on the GPU ("drop-in").

Run by tests/test_gpu_reference_dropin.py in a fresh process:  python tests/dropin_worker.py <reference dir> <soft_fp8 0|1>

What it does is INTEGRATION.md section 3's op-level edits, applied by monkeypatch instead of by editing the
reference tree (which is read-only and absent from the GPU box unless a copy is staged for this test):
  chitu_backend (pybind module, csrc/binding.cpp:11)        -> chitu_amd.chitu_backend
  chitu.ops.{apply_rotary_pos_emb, act_quant_deepseek_v3, weight_dequant_deepseek_v3,
             weight_dequant_soft_fp8_deepseek_v3, fp8_gemm_deepseek_v3, soft_fp8_gemm_deepseek_v3,
             append_to_paged_kv_cache}                       -> chitu_amd.ops (same names)
  chitu.fused_moe.{fused_experts, fused_experts_impl, moe_align_block_size, per_token_group_quant_fp8}
                                                             -> chitu_amd.fused_moe
  chitu.cache_manager.PagedKVCacheManager                    -> chitu_amd.cache_manager.PagedKVCacheManager
  attn backend (backend.py:259-270)                          -> chitu_amd.attn_backend.HipAttnBackend
  device name                                                -> "AMD Instinct MI355X" (not NVIDIA, not muxi)
Then the reference's unmodified TransformerDeepSeekV3 (mla_absorb="absorb-without-precomp", merged qkv / gate-up)
prefills the two ragged prompts of tests/golden/ref_model_v3.npz and decodes two steps, driven as the reference's
executor drives it (executor.py:118-148), and the logits are compared with
  (a) the fixture = the same model run by the reference on CPU (Triton interpreter; loose bar: its fp8 casts are the
      interpreter's defective ones, tests/test_oracle_golden.py), and
  (b) chitu_amd's own re-wired decoder on the same weights (tight bar: same kernels, fused differently).
soft_fp8 = 1 takes the reference's non-NVIDIA soft-fp8 branches: linears dequantise with weight_dequant_soft_fp8 and
run F.linear (model_deepseek_v3.py:85-98), the MoE dequantises the experts and calls fused_experts(use_fp8_w8a8=False)
(:975-993) -- the bf16 mode of chitu_amd.fused_moe.
Prints one JSON line.

DROPIN_TRACE=1 is the per-op watchdog asked for by VERDICT r05 item 1: every patched op, torch's F.linear and the model's
prefill / decode calls write "enter" (name, shapes, seconds since start) to stderr, flushed, BEFORE they run, and "leave"
with the elapsed time after a device synchronize; faulthandler dumps the Python stack every 30 s.  A hang therefore names the
launch that never retired, and a slow first call (library warm-up) shows up as one long "leave".
"""

import json
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


class AD(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


def _install_trace(torch, modules):
    """Wrap the named callables of `modules` (list of (module, [names])) with enter / leave lines on stderr."""
    import faulthandler
    import time

    faulthandler.dump_traceback_later(30, repeat=True, file=sys.stderr)
    t0 = time.time()
    seen = {}

    def describe(a):
        if isinstance(a, torch.Tensor):
            return f"{str(a.dtype).replace('torch.', '')}{list(a.shape)}"
        if isinstance(a, (list, tuple)) and len(a) < 6:
            return "[" + ",".join(describe(x) for x in a) + "]"
        return type(a).__name__ if not isinstance(a, (int, float, bool, str, type(None))) else repr(a)

    def wrap(label, fn):
        def traced(*a, **k):
            sig = label + "(" + ", ".join(describe(x) for x in a) + ")"
            first = sig not in seen
            seen[sig] = seen.get(sig, 0) + 1
            if first:
                sys.stderr.write(f"TRACE {time.time() - t0:8.3f} enter {sig}\n")
                sys.stderr.flush()
            t = time.time()
            out = fn(*a, **k)
            if first or time.time() - t > 0.5:
                torch.cuda.synchronize()
                sys.stderr.write(f"TRACE {time.time() - t0:8.3f} leave {label} {1e3 * (time.time() - t):.1f} ms\n")
                sys.stderr.flush()
            return out

        return traced

    for mod, names in modules:
        for n in names:
            setattr(mod, n, wrap(f"{mod.__name__.split('.')[-1]}.{n}", getattr(mod, n)))  # (a class: plain functions, self is a[0])
    return lambda msg: (sys.stderr.write(f"TRACE {time.time() - t0:8.3f} {msg}\n"), sys.stderr.flush())


def main(ref_dir: str, soft_fp8: bool):
    import numpy as np
    import torch

    sys.dont_write_bytecode = True
    sys.path.insert(0, ref_dir)
    import chitu_amd.attn_backend as a_attn
    import chitu_amd.cache_manager as a_cache
    import chitu_amd.chitu_backend as a_backend
    import chitu_amd.fused_moe as a_moe
    import chitu_amd.ops as a_ops

    note = lambda msg: None
    if os.environ.get("DROPIN_TRACE") == "1":
        import torch.nn.functional as F_

        note = _install_trace(torch, [
            (a_ops, ["apply_rotary_pos_emb", "act_quant_deepseek_v3", "weight_dequant_deepseek_v3", "weight_dequant_soft_fp8_deepseek_v3",
                     "fp8_gemm_deepseek_v3", "soft_fp8_gemm_deepseek_v3", "append_to_paged_kv_cache"]),
            (a_moe, ["fused_experts", "moe_align_block_size"]),
            (F_, ["linear", "silu", "embedding"]),
            (a_attn.HipAttnBackend, ["mla_attn_with_kvcache", "attn_varlen_func", "prepare_metadata_for_decode"]),
        ])
        note("imports done, tracing on")
    sys.modules["chitu_backend"] = a_backend
    for name in ("tiktoken", "tiktoken.load"):
        m = types.ModuleType(name)
        if name == "tiktoken.load":
            m.load_tiktoken_bpe = lambda *a, **k: {}
        sys.modules[name] = m
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("REF_MASTER_PORT", "29547"), RANK="0", WORLD_SIZE="1",
                      LOCAL_RANK="0")
    import torch.distributed as dist

    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    import chitu.device_type as dtmod

    dtmod._device_name = "AMD Instinct MI355X"
    from tests.golden.gen_ref_model import PROMPTS, TINY, fill

    import chitu.global_vars as gv

    models = AD(**TINY)
    infer = AD(tp_size=1, pp_size=1, max_reqs=4, cache_type="paged", soft_fp8=soft_fp8, use_cuda_graph=False, attn_type="hip",
               pp_layer_partition=None, max_seq_len=256, mla_absorb="absorb-without-precomp", op_impl="torch")
    gv.set_global_variables(AD(models=models, infer=infer))
    from chitu import tensor_parallel as rtp

    rtp.init_tp(1, 1)
    # ---- the op-level edits of INTEGRATION.md section 3
    import chitu.cache_manager as r_cache
    import chitu.fused_moe as r_moe
    import chitu.ops as r_ops

    patched = []
    for name in ("apply_rotary_pos_emb", "act_quant_deepseek_v3", "weight_dequant_deepseek_v3", "weight_dequant_soft_fp8_deepseek_v3",
                 "fp8_gemm_deepseek_v3", "soft_fp8_gemm_deepseek_v3", "append_to_paged_kv_cache"):
        setattr(r_ops, name, getattr(a_ops, name))
        patched.append("ops." + name)
    for name in ("fused_experts", "fused_experts_impl", "moe_align_block_size", "per_token_group_quant_fp8"):
        setattr(r_moe, name, getattr(a_moe, name))
        patched.append("fused_moe." + name)
    r_cache.PagedKVCacheManager = a_cache.PagedKVCacheManager
    import chitu.models.model as r_model
    import chitu.models.model_deepseek_v3 as rds

    assert rds.fused_experts is a_moe.fused_experts and rds.fp8_gemm_deepseek_v3 is a_ops.fp8_gemm_deepseek_v3
    assert rds.PagedKVCacheManager is a_cache.PagedKVCacheManager and r_model.PagedKVCacheManager is a_cache.PagedKVCacheManager
    from chitu.utils import VarLens

    torch.set_default_dtype(torch.bfloat16)  # the reference's Backend does this before building a model (backend.py:404)
    cache = a_cache.PagedKVCacheManager(0, models.n_layers, num_hot_req=4, block_size=64, max_seq_len=256, device="cuda",
                                        kv_shape_per_sample=(models.kv_lora_rank + models.qk_rope_head_dim,),
                                        dtype=torch.bfloat16)
    backend = a_attn.HipAttnBackend(local_n_heads=models.n_heads, kv_lora_rank=models.kv_lora_rank,
                                    qk_rope_head_dim=models.qk_rope_head_dim, qk_nope_head_dim=models.qk_nope_head_dim,
                                    max_seq_len=256)
    model = rds.TransformerDeepSeekV3(models, cache, max_position_embeddings=256, pipeline_parallel_size=1,
                                      model_parallel_size=1, attn_backend=backend, op_impl="torch",
                                      mla_absorb="absorb-without-precomp")
    named = sorted(model.named_parameters(), key=lambda kv: kv[0])
    fill(named)  # the fixture's weights: same seed, same sorted-name order, drawn on the CPU generator
    model = model.to("cuda")
    ids = ["a", "b"]
    g = np.load(os.path.join(HERE, "golden", "ref_model_v3.npz"))
    fed = torch.from_numpy(g["fed"])  # the tokens the reference fed (its own greedy picks): teacher forcing

    def run_reference_model(cache):
        """Prefill + two decode steps, driven as the reference's executor drives the model (executor.py:118-148)."""
        model.cache = cache
        for layer in model.layers:
            layer.attn.cache = cache
        vl = VarLens(PROMPTS, "cuda")
        cache.curr_varlens, cache.curr_req_ids = vl, ids
        note("prefill")
        res = [model.prefill(PROMPTS).float().clone()]
        cache.finalize_cache_all_prefill(ids, vl)
        for step in range(2):
            cache.prepare_cache_decode(ids)
            cache.prepare_block_table_for_decode(ids)
            note(f"decode step {step}")
            lg = model.decode(fed[step].view(-1, 1).cuda(), [cache.seq_lens[r] for r in ids]).view(len(ids), -1).float()
            cache.finalize_cache_single_decode(ids)
            res.append(lg.clone())
        torch.cuda.synchronize()
        note("run done")
        return res

    note("model built and on the device")
    outs = run_reference_model(cache)

    def rel(a, b):
        return ((a - b).abs().max() / b.abs().max()).item()

    res = {"soft_fp8": soft_fp8, "patched": patched, "device": torch.cuda.get_device_name(0)}
    # ---- (c) the same run with the MoE's fused_experts computed by the CPU oracle (test infrastructure): isolates
    # chitu_amd.fused_moe's mode of this run (fp8 W8A8, or the bf16 mode under soft_fp8) inside the reference's own forward
    from oracle import moe as omoe

    def oracle_fused_experts(hidden_states, w1, w2, topk_weights, topk_ids, inplace=False, use_fp8_w8a8=False,
                             w1_scale=None, w2_scale=None, **kw):
        c = lambda t: None if t is None else t.detach().cpu()
        if use_fp8_w8a8:
            y = omoe.fused_experts_fp8(c(hidden_states), c(w1), c(w2), c(topk_weights), c(topk_ids), c(w1_scale), c(w2_scale))
        else:
            y = omoe.fused_experts_bf16(c(hidden_states), c(w1), c(w2), c(topk_weights), c(topk_ids))
        y = y.to(hidden_states.device)
        return hidden_states.copy_(y) if inplace else y

    calls = []
    real = rds.fused_experts

    def counting(*a, **k):
        calls.append((bool(k.get("use_fp8_w8a8")), bool(k.get("soft_fp8")), str(a[1].dtype)))
        return real(*a, **k)

    rds.fused_experts = counting
    run_reference_model(a_cache.PagedKVCacheManager(0, models.n_layers, num_hot_req=4, block_size=64, max_seq_len=256, device="cuda",
                                                    kv_shape_per_sample=(576,), dtype=torch.bfloat16))
    res["fused_experts_calls"] = sorted(set(calls))
    note("re-run with the CPU oracle's fused_experts")
    rds.fused_experts = oracle_fused_experts
    oouts = run_reference_model(a_cache.PagedKVCacheManager(0, models.n_layers, num_hot_req=4, block_size=64, max_seq_len=256,
                                                            device="cuda", kv_shape_per_sample=(576,), dtype=torch.bfloat16))
    rds.fused_experts = real
    res["vs_same_run_with_oracle_moe"] = [rel(o, m) for o, m in zip(outs, oouts)]
    want = [torch.from_numpy(g[k]) for k in ("prefill", "d0", "d1")]
    res["vs_reference_cpu_run"] = [rel(o.cpu(), w) for o, w in zip(outs, want)]
    res["greedy_equal_reference"] = [bool((o.cpu().argmax(-1) == w.argmax(-1)).all()) for o, w in zip(outs, want)]
    res["reference_top2_margin_over_peak"] = [float(((w.topk(2).values[:, 0] - w.topk(2).values[:, 1]) / w.abs().max()).min()) for w in want]

    # ---- (b) chitu_amd's own decoder on the same weights
    note("chitu_amd decoder on the same weights")
    from chitu_amd.deepseek_v3 import DeepSeekV3Args, DeepSeekV3Decoder, refresh_derived_layouts
    from tests.util import ref_model_case

    _, params = ref_model_case()
    keys = ("vocab_size", "dim", "inter_dim", "moe_inter_dim", "n_layers", "n_dense_layers", "n_heads", "n_routed_experts",
            "n_shared_experts", "n_activated_experts", "n_expert_groups", "n_limited_groups", "route_scale", "score_func",
            "q_lora_rank", "kv_lora_rank", "qk_nope_head_dim", "qk_rope_head_dim", "v_head_dim", "rope_theta", "rope_factor")
    args = DeepSeekV3Args(**{k: TINY[k] for k in keys}, gate_bias=False, shard_degree=1)
    cache2 = a_cache.PagedKVCacheManager(0, args.n_layers, num_hot_req=4, block_size=64, max_seq_len=256, device="cuda",
                                         kv_shape_per_sample=(576,), dtype=torch.bfloat16)
    mine = DeepSeekV3Decoder(args, cache2, a_attn.HipAttnBackend(local_n_heads=16, max_seq_len=256), max_position_embeddings=256,
                             device="cuda")
    for n, p in mine.named_parameters():
        p.data.copy_(params[{"embed_weight": "embed.weight", "head_weight": "head.weight"}.get(n, n)])
    refresh_derived_layouts(mine)
    mouts = [mine.prefill(PROMPTS, ids).float().clone()]
    for step in range(2):
        cache2.prepare_cache_decode(ids)
        cache2.prepare_block_table_for_decode(ids)
        mouts.append(mine.decode(fed[step].cuda(), use_graph=False).float().clone())
        cache2.finalize_cache_single_decode(ids)
    torch.cuda.synchronize()
    res["vs_chitu_amd_decoder"] = [rel(o, m) for o, m in zip(outs, mouts)]
    res["greedy_equal_chitu_amd_decoder"] = [bool((o.argmax(-1) == m.argmax(-1)).all()) for o, m in zip(outs, mouts)]
    res["kv_pages_equal_chitu_amd_decoder"] = bool(all(
        (cache.get_paged_kv_cache(i).float() - cache2.get_paged_kv_cache(i).float()).abs().max().item() < 0.05 for i in range(args.n_layers)))
    print("DROPIN " + json.dumps(res), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] == "1")
