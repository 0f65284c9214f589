#!/usr/bin/env python3
"""How many distinct routed experts does a synthetic decode step hit per MoE layer?
(drives the algorithmic bytes of the step: a collapsed token stream under-counts the work)."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def probe(model, cache, bs, ctx, tokens_fn, tag, steps=3):
    from chitu_amd.deepseek_v3 import GateDeepSeekV3
    rec = []
    hooks = [m.register_forward_hook(lambda mod, i, o: rec.append(o[1].clone())) for m in model.modules() if isinstance(m, GateDeepSeekV3)]
    reqs = [f"{tag}{i}" for i in range(bs)]
    for r in reqs:
        cache.register_sequence(r, ctx)
    gen = torch.Generator(device="cuda").manual_seed(5)
    tokens = torch.randint(100, 1000, (bs,), device="cuda", generator=gen)
    out = []
    for s in range(steps):
        rec.clear()
        cache.prepare_cache_decode(reqs); cache.prepare_block_table_for_decode(reqs)
        logits = model.decode(tokens, use_graph=False)
        tokens = tokens_fn(logits.argmax(dim=-1), s)
        cache.finalize_cache_single_decode(reqs)
        d = [int(t[:, :8].unique().numel()) for t in rec]
        out.append((sum(d) / len(d), min(d), max(d), tokens.unique().numel()))
    for h in hooks: h.remove()
    for r in reqs: cache.finalize_cache_all_decode(r)
    return out

if __name__ == "__main__":
    ns = types.SimpleNamespace(layers=int(os.environ.get("LAYERS", "61")), ctx=1024, steps=8, warmup=2, bs=16)
    torch.cuda.set_device(0)
    margs, model, cache = bench.build_model(ns, 0)
    g = torch.Generator(device="cuda").manual_seed(9)
    print("greedy   :", probe(model, cache, 16, 1024, lambda t, s: t, "a"))
    # cosine similarity between different sequences' gate inputs, per layer
    from chitu_amd.deepseek_v3 import MoEDeepSeekV3
    sims = []
    def hk(mod, i, o):
        h = i[0].float(); h = h / h.norm(dim=-1, keepdim=True); s = h @ h.T
        sims.append(float((s.sum() - s.diag().sum()) / (s.numel() - s.shape[0])))
    hooks = [m.register_forward_hook(hk) for m in model.modules() if isinstance(m, MoEDeepSeekV3)]
    probe(model, cache, 16, 1024, lambda t, s: t, "c", steps=1)
    print("mean off-diagonal cos-sim of gate inputs per MoE layer:", [round(s, 2) for s in sims])
