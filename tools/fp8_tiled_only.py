"""Only the tiled fp8 GEMM at two R1 prefill shapes, a few launches each (a target for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chitu_amd import ops
gd = torch.Generator(device="cuda").manual_seed(5)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for name, (N, K) in {"wqkv_a": (2112, 7168), "wo": (7168, 2048), "dense_w1w3": (4608, 7168)}.items():
    x = torch.randn(T, K, device="cuda", generator=gd).to(torch.bfloat16)
    xq, xs = ops.act_quant_deepseek_v3(x)
    w = (torch.randn(N, K, device="cuda", generator=gd) * 0.5).to(torch.float8_e4m3fn)
    ws = torch.rand((N + 127) // 128, (K + 127) // 128, device="cuda", generator=gd) * 0.02 + 0.01
    for _ in range(6):
        ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
