#!/usr/bin/env python3
"""Debug: bench.py's in-process sequence with stage markers (main model -> extras)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def mark(s):
    torch.cuda.synchronize()
    print("[stage ok]", s, file=sys.stderr, flush=True)

sys.argv = ["bench.py"] + sys.argv[1:]
a = bench.parse()
torch.cuda.set_device(0)
margs, model, cache = bench.build_model(a, 0)
mark("build")
bench.measure(model, cache, a.bs, a.ctx, 8, 2, 1, True, "m"); mark("bs16")
bench.measure(model, cache, 1, a.ctx, 8, 2, 1, True, "s"); mark("bs1")
routing = bench.capture_step_routing(model, cache, a.bs, a.ctx); mark("routing")
roof = bench.roofline_dominant_kernel(model, routing, margs, a.bs); mark("roofline")
del model, cache
torch.cuda.empty_cache(); mark("freed")
extras = (("llama", bench.llama3_8b_extra), ("v2lite", bench.v2_lite_extra), ("mixtral", bench.mixtral_extra))
only = os.environ.get("BISECT_ONLY")
for name, fn in extras:
    if only and name not in only.split(","):
        continue
    fn(8, 2, a.ctx); mark(name)
print("all ok")
