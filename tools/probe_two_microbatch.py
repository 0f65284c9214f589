#!/usr/bin/env python3
"""Probe: does a bs-16 decode step get shorter as TWO micro-batches of 8 replayed on two streams?

The step is half byte streaming (the expert GEMMs, every CU busy) and half a chain of latency-bound launches that
use a few CUs each.  Two independent half-batches on two hardware queues could put one half's
chain under the other half's streaming -- at the price of streaming the experts both halves share twice (bs 16 touches
~102 distinct routed experts per layer, two bs-8 halves ~58 each).  Whether the hardware interleaves two queues that
way is the question; this measures it:

    python tools/probe_two_microbatch.py [layers=61] [steps=32]

  one engine, bs 16, one graph                       (the shipped step)
  one engine, bs 8, one graph                        (a half alone)
  two engines (bs 8 + bs 8), weights shared, a graph each, replayed on two streams, one host thread
  the same two engines on ONE stream                 (no overlap possible: the control)
Prints one JSON object.  Results of a half are those of the same sequences in the full batch (the kernels are
batch-invariant, tests/test_gpu_deepseek.py), so this is a scheduling experiment, not an approximation.
"""
import json
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from chitu_amd import sampling, workspace
from chitu_amd.attn_backend import HipAttnBackend
from chitu_amd.cache_manager import PagedKVCacheManager
from chitu_amd.deepseek_v3 import DeepSeekV3Decoder
from chitu_amd.xgmi import XgmiComm

torch.cuda.set_device(0)
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 61
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ctx, warm = 1024, 4
ns = SimpleNamespace(layers=layers, ctx=ctx, steps=steps, warmup=warm, bs=16, no_bs1=True, router_std=None)
margs, model16, cache16 = bench.build_model(ns, 0)


def side_by_side_streams(n):
    """n streams that really run side by side (two fresh HIP streams can share a hardware queue): each admitted by a
    tiny all-reduce among the chosen ones, which only completes if they run concurrently (tests/test_gpu_xgmi.py)."""
    chosen = []
    for _ in range(16):
        if len(chosen) == n:
            break
        cand = chosen + [torch.cuda.Stream()]
        if len(cand) == 1:
            chosen = cand
            continue
        comms = [XgmiComm(r, len(cand), max_rows=1, max_dim=64, timeout_ms=200) for r in range(len(cand))]
        XgmiComm.connect_local(comms)
        part = torch.ones(1, 64, dtype=torch.bfloat16, device="cuda")
        torch.cuda.synchronize()
        for _ in range(2):
            for r, s in enumerate(cand):
                with torch.cuda.stream(s):
                    comms[r].allreduce_rmsnorm(part)
        torch.cuda.synchronize()
        ok = all(c.status() == 0 for c in comms)
        for c in comms:
            c.close()
        if ok:
            chosen = cand
    return chosen


class Engine:
    def __init__(self, tag, bs, share_from):
        max_seq = ctx + 2 * (steps + warm) + 256
        self.cache = PagedKVCacheManager(0, margs.n_layers, num_hot_req=bs, block_size=64, max_seq_len=max_seq, device="cuda",
                                         kv_shape_per_sample=(margs.kv_lora_rank + margs.qk_rope_head_dim,), dtype=torch.bfloat16)
        self.cache.paged_kv_cache.normal_(0, 0.5)
        be = HipAttnBackend(local_n_heads=margs.n_heads // bench.SHARD, max_seq_len=max_seq)
        self.model = DeepSeekV3Decoder(margs, self.cache, be, max_position_embeddings=max(max_seq, 4097), device="cuda")
        for (na, pa), (nb, pb) in zip(share_from.named_parameters(), self.model.named_parameters()):
            assert na == nb and pa.shape == pb.shape
            pb.data = pa.data  # the same weights in HBM: what two micro-batches of one model read
        torch.cuda.empty_cache()
        self.tag, self.bs = tag, bs
        self.reqs = [f"{tag}{i}" for i in range(bs)]
        for r in self.reqs:
            self.cache.register_sequence(r, ctx)
        self.tokens = torch.randint(100, 1000, (bs,), device="cuda")

    def step(self):
        workspace.set_namespace(self.tag)
        self.cache.prepare_cache_decode(self.reqs)
        self.cache.prepare_block_table_for_decode(self.reqs)
        logits = self.model.decode(self.tokens, use_graph=True)
        self.tokens = sampling.argmax(logits)
        self.cache.finalize_cache_single_decode(self.reqs)


def timed(engines, streams, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for e, s in zip(engines, streams):
            with torch.cuda.stream(s):
                e.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


res = {"layers": layers, "steps": steps, "ctx": ctx}
workspace.set_namespace(None)
res["bs16_one_graph_ms"] = round(bench.measure(model16, cache16, 16, ctx, steps, warm, 1, True, "full") * 1e3 / steps, 4)
res["bs8_one_graph_ms"] = round(bench.measure(model16, cache16, 8, ctx, steps, warm, 1, True, "half") * 1e3 / steps, 4)
streams = side_by_side_streams(2)
res["side_by_side_streams"] = len(streams)
a, b = Engine("mbA", 8, model16), Engine("mbB", 8, model16)
if len(streams) == 2:
    timed([a, b], streams, warm)  # capture + warm
    res["two_bs8_two_streams_ms"] = round(timed([a, b], streams, steps), 4)
one = [streams[0], streams[0]]
timed([a, b], one, warm)
res["two_bs8_one_stream_ms"] = round(timed([a, b], one, steps), 4)
if len(streams) == 2:
    res["two_bs8_two_streams_again_ms"] = round(timed([a, b], streams, steps), 4)
workspace.set_namespace(None)
print(json.dumps(res))
