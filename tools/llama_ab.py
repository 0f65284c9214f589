#!/usr/bin/env python3
"""Step-time A/B of Llama-3-8B decode variants in ONE process on ONE box (model built once, a fresh hipGraph per variant):
    python tools/llama_ab.py [--steps 40] [--reps 2] [--bs 1,2,4]
Variants: the two add + RMSNorm launches of a layer as their own kernels (fuse 0) vs as the prologue of the GEMM behind
them (ops.bf16_linear_add_norm / bf16_linear_silu_add_norm), and any launch-variant overrides given as
--opt name=value (chitu_amd._lib.DEBUG_OPTIONS), each measured with and without the fusion."""
import argparse
import contextlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from chitu_amd import _lib, llama  # noqa: E402


@torch.inference_mode()
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--bs", default="1")
    ap.add_argument("--opt", action="append", default=[], help="name=value launch-variant override, measured as its own variant")
    a = ap.parse_args()
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager

    args = llama.LlamaArgs()
    max_seq = a.ctx + (a.steps + a.warmup) * 64 + 512
    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=16, block_size=256, max_seq_len=max_seq, device="cuda",
                                n_local_kv_heads=args.n_kv_heads, head_dim=args.head_dim, dtype=torch.bfloat16)
    model = llama.LlamaDecoder(args, cache, HipAttnBackend(local_n_heads=args.n_heads, max_seq_len=max_seq),
                               max_position_embeddings=max_seq, device="cuda")
    llama.init_synthetic_(model, seed=3)
    cache.paged_k_cache.normal_(0, 0.5)
    cache.paged_v_cache.normal_(0, 0.5)
    variants = [("norm launches", 0, None), ("norm in GEMM prologue", 4, None)]
    for o in a.opt:
        name, val = o.split("=")
        variants += [(f"{o}, norm launches", 0, (name, int(val))), (f"{o}, norm in GEMM prologue", 4, (name, int(val)))]
    n = 0
    for bs in [int(b) for b in a.bs.split(",")]:
        for rep in range(a.reps):
            for label, fuse, opt in variants:
                llama.FUSE_NORM_MAX_BS = fuse
                model.graphs.clear()
                ctxm = _lib.debug_option(*opt) if opt else contextlib.nullcontext()
                with ctxm:
                    n += 1
                    dt = bench.measure(model, cache, bs, a.ctx, a.steps, a.warmup, 1, True, f"v{n}_")
                print(json.dumps({"bs": bs, "rep": rep, "variant": label, "ms_per_step": round(dt / a.steps * 1e3, 4)}), flush=True)


if __name__ == "__main__":
    main()
