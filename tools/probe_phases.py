#!/usr/bin/env python3
"""Phase timing INSIDE the latency-bound kernels (probe build: every .hip compiled with -DCHITU_PROBE, see common.h):
    make -C chitu_amd/csrc probe            -> build_probe/libchitu_hip_probe.so
    CHITU_HIP_LIB=build_probe/libchitu_hip_probe.so python tools/probe_phases.py [bs=16]
Runs eager decode steps of an 8-layer R1 rank shard (so the kernels see the step's real inputs and a cold cache) and
prints, for workgroup 0 of the LAST launch of each probed kernel, the 100 MHz wall-clock deltas between its marks."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from chitu_amd import _lib  # noqa: E402

NAMES = {
    ("gate", "gate_route_align_wg_kernel"): [(0, "start"), (1, "logits loaded + summed"), (5, "scores (sigmoid, bias, keys)"),
                                              (6, "group selection"), (7, "top-k rounds"), (2, "weights normalised, ids in LDS"),
                                              (3, "workgroup barrier (slowest token wave)"), (20, "sort: sentinels + zeroing"),
                                              (21, "sort: counts"), (22, "sort: scan + expert ids"), (4, "sort: scatter; done")],
    ("fp8_gemm", "fp8_gemm_kernel<1,8,DEEP> (wqkv_a)"): [(0, "start"), (1, "all loads issued"), (2, "first K block multiplied"),
                                                          (3, "K loop done"), (4, "LDS reduce + stores issued")],
    ("absorb", "mla_merge_uv_quant_kernel"): [(0, "start"), (1, "merged row written to LDS"), (2, "barrier"), (3, "W_UV projection done"),
                                              (4, "group scale + quotients")],
    ("norm", "rmsnorm_add_kernel<group quant, 1 term> (ffn_norm)"): [(0, "start"), (1, "inputs arrived, residual added"), (2, "norm + quant + stores issued")],
    ("mla_q_proj", "mla_q_proj_kernel<1> (first GEMM workgroup)"): [(0, "start"), (1, "all loads issued"), (2, "activation rows arrived, partial mean squares in LDS"),
                                                                    (3, "barrier, row sums, rsqrt"), (4, "first K block: norm + quant + MFMAs (weights arrived)"),
                                                                    (5, "all K blocks"), (6, "LDS reduce + stores issued")],
    ("mla_q_proj", "mla_q_proj_kernel<1> (first KV workgroup)"): [(10, "start"), (11, "page row written")],
    ("mla_decode", "mla_decode_kernel"): [(8, "start"), (9, "seqlens loaded"), (10, "first KV tile staged in LDS"), (11, "QK^T done"),
                          (12, "tile loop done (softmax, PV)"), (13, "partials stored")],
}


@torch.inference_mode()
def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 16

    class A:
        layers, ctx, steps, warmup, bs, no_bs1, router_std = 8, 1024, 4, 2, 16, True, None

    A.bs = bs
    margs, model, cache = bench.build_model(A, 0)
    reqs = [f"p{i}" for i in range(bs)]
    for r in reqs:
        cache.register_sequence(r, 1024)
    tokens = torch.randint(100, 1000, (bs,), device="cuda")
    for _ in range(3):
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        tokens = model.decode(tokens, use_graph=False).argmax(-1)
        cache.finalize_cache_single_decode(reqs)
    for (tu, k), marks in NAMES.items():
        buf = (ctypes.c_ulonglong * 32)()
        rc = getattr(_lib.lib(), f"chitu_hip_probe_read_{tu}")(buf)
        assert rc == 0, rc
        t0 = buf[marks[0][0]]
        print(k)
        prev = t0
        for idx, name in marks[1:]:
            t = buf[idx]
            print(f"   +{(t - prev) * 0.01:6.2f} us  -> {name}   (at {(t - t0) * 0.01:6.2f} us)")
            prev = t


if __name__ == "__main__":
    main()
