"""How many elements of fused_experts (fp8 W8A8) fall outside the element-wise bar vs the oracle, per test shape?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import moe as omoe
from tests.test_gpu_moe import make_case, run_hip, REL_TOL

SHAPES = [(1, 32, 8, 7168, 256), (16, 32, 8, 7168, 256), (5, 8, 2, 256, 128), (7, 8, 3, 384, 640), (4, 64, 6, 2048, 384),
          (16, 64, 8, 2048, 1408), (5, 8, 2, 512, 2048), (40, 4, 2, 256, 1024)]
for (M, E, topk, K, I) in SHAPES:
    args = make_case(M, E, topk, K, I, seed=M * 1000 + E)
    out = run_hip(*args).float()
    x, w1, w2, w1s, w2s, ids, wts = args
    ref = omoe.fused_experts_fp8(x, w1, w2, wts, ids, w1s, w2s).float()
    peak = ref.abs().max()
    bound = 1e-2 * ref.abs() + 0.5 * REL_TOL * peak
    bad = (out - ref).abs() > bound
    print((M, E, topk, K, I), "outside:", int(bad.sum()), "of", bad.numel(), "max err/peak", ((out - ref).abs().max() / peak).item(),
          "worst excess/peak", (((out - ref).abs() - bound).max() / peak).item(), flush=True)
