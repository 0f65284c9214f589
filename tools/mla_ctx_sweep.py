#!/usr/bin/env python3
"""MLA paged decode (chitu_hip_mla_decode + the fused merge / W_UV / quant launch) over the context length:
python tools/mla_ctx_sweep.py [bs=16] -> JSON lines {ctx, splits, decode_us, merge_us, kv_MB, decode_TBs, total_TBs}.
16 local heads (TP=8 rank of R1), 64-token pages, random latent cache; each launch pair is captured 50x in a
hipGraph and timed with events on the replay stream; kv bytes = bs * ctx * 576 * 2."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chitu_amd import ops  # noqa: E402
from chitu_amd.attn_backend import HipAttnBackend  # noqa: E402


@torch.inference_mode()
def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    H, C, R = 16, 512, 64
    g = torch.Generator(device="cuda").manual_seed(0)
    w_uv = (torch.randn(H, 128, C, device="cuda", generator=g) * 0.5).to(torch.float8_e4m3fn)
    sc = torch.rand(H * 2, C // 128, device="cuda", generator=g) * 0.02 + 0.01
    out = []
    for ctx in (1024, 4096, 8192, 32768):
        pages_per = ctx // 64 + 1
        cache = (torch.randn(bs * pages_per, 64, C + R, device="cuda", generator=g) * 0.5).to(torch.bfloat16)
        table = torch.randperm(bs * pages_per, device="cuda", generator=g).to(torch.int32).view(bs, pages_per)
        lens = torch.full((bs,), ctx, dtype=torch.int32, device="cuda")
        be = HipAttnBackend(local_n_heads=H, max_seq_len=ctx + 64)
        q_nope = torch.randn(bs, H, C, device="cuda", generator=g).to(torch.bfloat16)
        q_pe = torch.randn(bs, H, R, device="cuda", generator=g).to(torch.bfloat16)

        def attn():
            return be.mla_decode(q_nope, q_pe, cache, lens, table, 0.1, return_partials=True)

        def both():
            o = attn()
            if isinstance(o, tuple):
                return ops.mla_merge_absorb_uv_quant_fp8(o[0], o[1], bs, w_uv, sc, 4, 8, 1)
            return ops.absorb_uv_quant_fp8(o, w_uv, sc, 4, 8, 1)

        res = {}
        o = attn()
        splits = o[1] if isinstance(o, tuple) else 1
        for name, fn in (("decode", attn), ("both", both)):
            fn()
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                for _ in range(50):
                    fn()
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) * 1e3 / 50
            del gr
        kv = bs * ctx * (C + R) * 2
        row = {"bs": bs, "ctx": ctx, "splits": splits, "decode_us": round(res["decode"], 2),
               "merge_uv_quant_us": round(res["both"] - res["decode"], 2), "kv_MB": round(kv / 1e6, 1),
               "decode_TBs": round(kv / res["decode"] / 1e6, 3), "decode_plus_merge_TBs": round(kv / res["both"] / 1e6, 3),
               "frac_of_8TBs": round(kv / res["decode"] / 1e6 / 8, 3)}
        print(json.dumps(row), flush=True)
        out.append(row)
        del cache
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
