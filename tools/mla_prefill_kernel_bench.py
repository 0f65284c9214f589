#!/usr/bin/env python3
"""MLA prefill attention kernels alone (no model): parity of the flash kernel against the bit-exact one and the oracle on a ragged
batch, then time per launch at T tokens x 16 heads (one R1 rank) with HIP events: python tools/mla_prefill_kernel_bench.py [T ...]
FLOP count: causal, 16 heads x (576 + 512) MACs per (query, key) pair."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(be, mode, q, kv, cu, mx, scale=0.1352):
    os.environ["CHITU_MLA_PREFILL"] = mode
    return be.attn_varlen_func(q, kv, kv[..., :512], cu, cu, mx, mx, causal=True, softmax_scale=scale)


@torch.inference_mode()
def main():
    from chitu_amd.attn_backend import HipAttnBackend
    from oracle import mla as omla

    be = HipAttnBackend(local_n_heads=16)
    g = torch.Generator().manual_seed(31)
    if os.environ.get("PARITY", "1") == "1":
        for H, seqs in ((16, [1, 2, 3, 4, 5, 7, 8, 9, 63, 64, 65, 66, 127, 128, 129, 300]), (32, [7, 200]), (8, [70]), (16, [700])):
            T = sum(seqs)
            cu = torch.tensor([0] + list(np.cumsum(seqs)), dtype=torch.int32).cuda()
            q = (torch.randn(T, H, 576, generator=g) * 0.3).to(torch.bfloat16).cuda()
            kv = torch.randn(T, 1, 576, generator=g).to(torch.bfloat16).cuda()
            b = HipAttnBackend(local_n_heads=H)
            a = run(b, "exact", q, kv, cu, max(seqs)).float().cpu()
            f = run(b, "flash", q, kv, cu, max(seqs)).float().cpu()
            ref = omla.mla_prefill(q.cpu(), kv.cpu()[:, 0], cu.cpu(), 0.1352).float()
            pk = ref.abs().max().item()
            print(json.dumps({"H": H, "seqs": seqs, "finite": bool(torch.isfinite(f).all()),
                              "flash_vs_exact": (f - a).abs().max().item() / pk, "flash_vs_oracle": (f - ref).abs().max().item() / pk,
                              "exact_vs_oracle": (a - ref).abs().max().item() / pk}), flush=True)
            if not torch.isfinite(f).all() or (f - ref).abs().max().item() / pk > 1e-2:
                bad = ((f - ref).abs().amax(dim=(1, 2)) / pk > 1e-2).nonzero().flatten().tolist()
                print("  bad tokens:", bad[:40], flush=True)
    for T in [int(a) for a in sys.argv[1:]] or [512, 2048, 4096]:
        cu = torch.tensor([0, T], dtype=torch.int32).cuda()
        q = (torch.randn(T, 16, 576, generator=g) * 0.3).to(torch.bfloat16).cuda()
        kv = torch.randn(T, 1, 576, generator=g).to(torch.bfloat16).cuda()
        flop = 2 * 16 * (T * (T + 1) / 2) * (576 + 512)
        for mode in ("exact", "flash"):
            for _ in range(3):
                run(be, mode, q, kv, cu, T)
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
            n = 20
            ev[0].record()
            for _ in range(n):
                run(be, mode, q, kv, cu, T)
            ev[1].record()
            torch.cuda.synchronize()
            us = ev[0].elapsed_time(ev[1]) * 1e3 / n
            print(json.dumps({"T": T, "mode": mode, "us": round(us, 1), "TFLOPs": round(flop / us / 1e6, 1), "frac_2.5PF": round(flop / us / 1e6 / 2500, 3)}), flush=True)


if __name__ == "__main__":
    main()
