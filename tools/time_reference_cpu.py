#!/usr/bin/env python3
"""Time the REFERENCE'S OWN decode on this host's CPU cores (BASELINE.md 4 / configs[0]; build container only:
needs /root/reference).  Nothing here is product code; the result is committed as
profiles/r02_ref_cpu_decode.json and quoted by bench.py's cpu_baseline beside the CPU port it times live.

  python tools/time_reference_cpu.py llama2_7b        BASELINE config 1 as stated: the reference's TransformerLlama,
        Llama-2-7B shapes (32 layers, dim 4096, 32 heads, vocab 32000), bf16, RefAttnBackend + contiguous KV cache,
        random weights, bs 1, 7-token prompt, 64 greedy decode steps.
  python tools/time_reference_cpu.py deepseek_r1_rank  the reference's TransformerDeepSeekV3 at the per-rank shapes
        of R1 under TP=8 (dim 7168, 16 heads, dense width 2304, 256 + 1 experts of width 256, vocab 16160), ONE
        dense + ONE MoE layer, bs 16, a 4-token prefill and then timed decode steps.  Its FP8 linears are the
        reference's Triton kernels, which on a CPU only run under TRITON_INTERPRET=1 (the reference has no other
        CPU path for FP8); the MoE takes its per-expert loop (model_deepseek_v3.py:1012-1061).
Shims: tests/golden/ref_shims.py."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
OUT = os.path.join(ROOT, "profiles", "r02_ref_cpu_decode.json")


def save(key, val):
    d = json.load(open(OUT)) if os.path.exists(OUT) else {}
    d[key] = val
    d["host"] = {"cores": os.cpu_count(), "where": "build container (no GPU)"}
    json.dump(d, open(OUT, "w"), indent=1, sort_keys=True)
    print(key, json.dumps(val))


def llama2_7b():
    import gen_ref_llama as g
    import torch

    g.TINY = dict(name="llama2-7b", type="llama", dim=4096, n_layers=32, n_heads=32, n_kv_heads=32, vocab_size=32000,
                  multiple_of=256, ffn_dim_multiplier=None, norm_eps=1e-5, rope_theta=10000.0)
    steps, times = 64, []
    orig_savez = g.np.savez_compressed
    g.np.savez_compressed = lambda *a, **k: None  # timing run: no fixture
    # instrument the decode loop: time.perf_counter around model.decode via a wrapper on the class
    import chitu.models.model as rmodel

    real_decode = rmodel.Transformer.decode

    def timed(self, *a, **k):
        t0 = time.perf_counter()
        r = real_decode(self, *a, **k)
        times.append(time.perf_counter() - t0)
        return r

    rmodel.Transformer.decode = timed
    t0 = time.perf_counter()
    g.NEW_TOKENS = steps
    g.main()
    total = time.perf_counter() - t0
    g.np.savez_compressed = orig_savez
    ms = sorted(times)
    save("llama2_7b_bf16_bs1", {
        "what": "reference TransformerLlama.decode on CPU, Llama-2-7B shapes, bf16, bs 1, 64 decode steps after a 7-token prefill",
        "threads": torch.get_num_threads(), "decode_steps": len(times), "ms_per_token_median": round(ms[len(ms) // 2] * 1e3, 1),
        "ms_per_token_mean": round(sum(ms) / len(ms) * 1e3, 1), "tok_s": round(len(ms) / sum(ms), 3),
        "wall_s_incl_build_and_prefill": round(total, 1)})


def deepseek_r1_rank(bs=16, steps=2):
    import gen_ref_model as g
    import torch

    cfg = dict(g.TINY, name="r1-rank", vocab_size=16160, dim=7168, inter_dim=2304, moe_inter_dim=256, n_layers=2,
               n_dense_layers=1, n_heads=16, n_routed_experts=256, n_shared_experts=1, n_activated_experts=8,
               n_expert_groups=8, n_limited_groups=4, q_lora_rank=1536)
    model, cache, rds = g.build_reference_model(cfg, max_seq_len=64, max_reqs=bs)
    g.fill(sorted(model.named_parameters(), key=lambda kv: kv[0]))
    from chitu.utils import VarLens

    prompts = [[5 + i, 17, 900, 33] for i in range(bs)]
    ids = [f"r{i}" for i in range(bs)]
    vl = VarLens(prompts, "cpu")
    cache.curr_varlens, cache.curr_req_ids = vl, ids
    t0 = time.perf_counter()
    with torch.inference_mode():
        lg = model.prefill(prompts)
    cache.finalize_cache_all_prefill(ids, vl)
    prefill_s = time.perf_counter() - t0
    tok = lg.float().argmax(-1)
    times = []
    for _ in range(steps):
        cache.prepare_cache_decode(ids)
        t0 = time.perf_counter()
        with torch.inference_mode():
            lg = model.decode(tok.view(-1, 1), [cache.seq_lens[r] for r in ids])
        times.append(time.perf_counter() - t0)
        cache.finalize_cache_single_decode(ids)
        tok = lg.view(bs, -1).float().argmax(-1)
    per_step = min(times)
    save(f"deepseek_r1_tp8_rank_bs{bs}", {
        "what": "reference TransformerDeepSeekV3.decode on CPU: per-rank R1 shapes (TP=8), 1 dense + 1 MoE layer, all 257 "
                "experts materialised, FP8 linears = the reference's Triton kernels under TRITON_INTERPRET=1, MoE = its "
                "per-expert loop, context 4-5 tokens",
        "threads": torch.get_num_threads(), "bs": bs, "layers": 2, "decode_steps_timed": steps,
        "s_per_step_2_layers": round(per_step, 2), "s_per_layer": round(per_step / 2, 2),
        "prefill_s_4_tokens_each": round(prefill_s, 1),
        "extrapolated_s_per_61_layer_step": round(per_step / 2 * 61, 1)})


if __name__ == "__main__":
    {"llama2_7b": llama2_7b, "deepseek_r1_rank": deepseek_r1_rank}[sys.argv[1]](*[int(v) for v in sys.argv[2:]])
