"""Micro-benchmark: cost of a dependent chain of tiny kernels (eager stream vs hipGraph replay)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chitu_amd import ops

torch.cuda.set_device(0)
q = torch.randn(1, 16, 64, device="cuda", dtype=torch.bfloat16)
k = torch.randn(1, 64, device="cuda", dtype=torch.bfloat16)
cos = torch.randn(1, 32, device="cuda"); sin = torch.randn(1, 32, device="cuda")
x = torch.randn(16, 7168, device="cuda", dtype=torch.bfloat16)
N = 1000

def chain_hip():
    a, b = q, k
    for _ in range(N):
        a, b = ops.apply_rotary_pos_emb(a, b, cos, sin, "llama")
    return a

def chain_torch():
    y = x
    for _ in range(N):
        y = y + 1
    return y

for name, fn in (("hip rope (ctypes)", chain_hip), ("torch add", chain_torch)):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"{name}: eager {dt / N * 1e6:.2f} us/kernel")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
    print(f"{name}: graph {dt / N * 1e6:.2f} us/kernel")
