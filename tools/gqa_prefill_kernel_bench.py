#!/usr/bin/env python3
"""GQA causal prefill attention alone (Llama-3-8B: 32 q heads, 8 kv heads, head_dim 128): the flash kernel against the
round-2 decode composition, us per launch with HIP events: python tools/gqa_prefill_kernel_bench.py [T ...]
FLOPs: causal, 32 heads x (128 + 128) MACs per (query, key) pair."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@torch.inference_mode()
def main():
    from chitu_amd.attn_backend import HipAttnBackend

    be = HipAttnBackend(local_n_heads=32)
    g = torch.Generator(device="cuda").manual_seed(3)
    for T in [int(a) for a in sys.argv[1:]] or [512, 2048, 8192]:
        cu = torch.tensor([0, T], dtype=torch.int32, device="cuda")
        q = (torch.randn(T, 32, 128, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        k = torch.randn(T, 8, 128, device="cuda", generator=g).to(torch.bfloat16)
        v = torch.randn(T, 8, 128, device="cuda", generator=g).to(torch.bfloat16)
        flop = 2.0 * 32 * (T * (T + 1) / 2) * 256
        for mode in ("flash", "compose"):
            if mode == "compose" and T > 2048:
                continue
            os.environ["CHITU_GQA_PREFILL"] = mode
            fn = lambda: be.attn_varlen_func(q, k, v, cu, cu, T, T, causal=True)  # noqa: E731
            for _ in range(2):
                fn()
            n = 10 if mode == "flash" else 2
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / n
            print(json.dumps({"T": T, "mode": mode, "us": round(us, 1), "TFLOPs": round(flop / us * 1e-6, 1),
                              "frac_2.5PF": round(flop / us * 1e-6 / 2500, 4)}), flush=True)


if __name__ == "__main__":
    main()
