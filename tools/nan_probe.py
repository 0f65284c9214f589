#!/usr/bin/env python3
"""Find the first module whose output is non-finite in a full-size synthetic decode step."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ns = types.SimpleNamespace(layers=int(os.environ.get("LAYERS", "8")), ctx=1024, steps=8, warmup=2, bs=16)
torch.cuda.set_device(0)
margs, model, cache = bench.build_model(ns, 0)
def stat(t):
    t = t.float()
    return f"finite={bool(torch.isfinite(t).all())} rms={float(t[torch.isfinite(t)].pow(2).mean().sqrt()):.3g} max={float(t[torch.isfinite(t)].abs().max()):.3g}"
def hk(name):
    def f(mod, i, o):
        outs = o if isinstance(o, (tuple, list)) else (o,)
        for k, t in enumerate(outs):
            if torch.is_tensor(t) and t.is_floating_point() and t.dtype in (torch.bfloat16, torch.float32, torch.float16):
                print(f"{name}[{k}] {tuple(t.shape)} {stat(t)}")
    return f
for n, m in model.named_modules():
    if n and n.count(".") <= 2:
        m.register_forward_hook(hk(n))
reqs = [f"r{i}" for i in range(16)]
for r in reqs: cache.register_sequence(r, 1024)
tokens = torch.randint(100, 1000, (16,), device="cuda")
cache.prepare_cache_decode(reqs); cache.prepare_block_table_for_decode(reqs)
logits = model.decode(tokens, use_graph=False)
print("logits", stat(logits))
