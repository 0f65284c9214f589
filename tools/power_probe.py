#!/usr/bin/env python3
"""Does the decode step run clock/power limited?  Replays the bs=16 graph for a few seconds while
sampling rocm-smi (sclk, power), once with the balanced synthetic router and once with a collapsed
one (all tokens pick similar experts => fewer expert bytes per step)."""
import os, subprocess, sys, threading, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True, timeout=5).stdout
            out.append(r.strip().splitlines()[-1])
        except Exception as e:
            out.append(repr(e))
        time.sleep(0.3)

ns = types.SimpleNamespace(layers=61, ctx=1024, steps=8, warmup=2, bs=16, no_bs1=True, router_std=None)
torch.cuda.set_device(0)
margs, model, cache = bench.build_model(ns, 0)
hdr = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()[0]
print(hdr)
for tag, scale in (("balanced", 1.0), ("collapsed", 4.2), ("balanced2", 1 / 4.2)):
    for n, p in model.named_parameters():
        if n.endswith("gate.weight"):
            p.data.mul_(scale)
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out)); th.start()
    dt = bench.measure(model, cache, 16, 1024, 250, 8, 1, True, tag)
    stop.set(); th.join()
    print(tag, f"{dt/250*1e3:.3f} ms/step")
    for l in out[:: max(1, len(out) // 6)]:
        print("   ", l)
