#!/usr/bin/env python3
"""Prefill timing of one TP=8 rank shard of DeepSeek-R1 FP8: python tools/prefill_bench.py [layers=8] [T ...]
One prompt of T tokens through `DeepSeekV3Decoder.prefill` (eager launches: absorb-mode MLA prefill kernel, fp8 GEMMs
that stream the weights once per 64 rows, fused MoE over T * 8 slots), T in {128, 512, 2048}; ms per layer and the
rank's tokens/s extrapolated to 61 layers.  Decode is the metric; this records where the step before it stands."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@torch.inference_mode()
def main():
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Args, DeepSeekV3Decoder, init_synthetic_

    layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    args = DeepSeekV3Args(shard_degree=8, n_layers=layers)
    cache = PagedKVCacheManager(0, layers, num_hot_req=2, block_size=64, max_seq_len=4096, device="cuda",
                                kv_shape_per_sample=(576,), dtype=torch.bfloat16)
    model = DeepSeekV3Decoder(args, cache, HipAttnBackend(local_n_heads=16, max_seq_len=4096), max_position_embeddings=4097,
                              device="cuda")
    init_synthetic_(model, seed=1)
    g = torch.Generator().manual_seed(0)
    for T in [int(a) for a in sys.argv[2:]] or (128, 512, 2048):
        prompt = torch.randint(100, 1000, (T,), generator=g).tolist()
        times = []
        for rep in range(3):
            rid = f"p{T}_{rep}"
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            model.prefill([prompt], [rid])
            torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
            cache.finalize_cache_all_decode(rid)
        best = min(times[1:])
        print(json.dumps({"prompt_tokens": T, "layers": layers, "ms": round(best * 1e3, 2), "ms_per_layer": round(best * 1e3 / layers, 3),
                          "rank_tok_s_at_61_layers": round(T / (best / layers * 61), 1)}), flush=True)


if __name__ == "__main__":
    main()
