#!/usr/bin/env python3
"""Decode-throughput benchmark of the MI355X-native hot path (BASELINE.json metric).

Workload (named in `config.workload`): DeepSeek-R1-671B FP8, one TP=8 rank shard per GPU -- all
61 layers, 16 local heads, all 257 experts at 1/8 width, replicated wqkv_a / gate / latent KV
cache -- i.e. exactly the bytes and kernels one GPU of the 8-GPU node executes per decode step
(35.7 GB/step at bs=16, 5.7 GB at bs=1).  Random synthetic weights and KV
(no network for checkpoints), batch `--bs` sequences at context `--ctx`, greedy sampling, the
whole step replayed as one hipGraph.  With N ranks live the step's 124 collectives (2 all-reduces
per layer, the embedding's, the logits all-gather) run across those N ranks: by default as the in-graph
xGMI kernels of chitu_amd/csrc/comm.hip (the step stays ONE hipGraph and every per-layer all-reduce is
fused into the norm launch that consumes it; `collective_transport` in the output says which path ran),
or -- CHITU_ALLREDUCE=rccl, or when the xGMI self-test fails -- through RCCL with the step replayed as
hipGraph pieces between the library calls (chitu_amd/graphs.py).  At N=8 this IS the metric's configuration.

`python bench.py --gpus N` with N > 1 and no launcher environment re-executes itself under
`torch.distributed.run` (one rank per GPU; when the box has fewer than N GPUs every rank shares cuda:0,
the process group is gloo, the layer count is cut to what fits, and the line is flagged invalid: a
functional check of the N > 1 code path, not a measurement).

A "step" = one decode token for the whole batch.  value = (N/8) * bs * K / T: N GPUs carry N/8 of the
model's weights, so N/8 of an 8-GPU node's tokens are attributed to them (weak scaling: per-GPU work fixed).
`node_tok_s` = bs*K/T is what an 8-GPU node emits at this per-rank step time; at N < 8 the collectives it
would pay span only the live ranks (`collectives_in_step` is 0 at N=1), so only N=8 is the metric itself.

`roofline` (dominant kernel, the routed experts' GEMM1 + SiLU): `achieved` = algorithmic bytes of one launch / its duration,
measured live with HIP events around replays of one hipGraph holding every MoE layer's launch back to back
(`avg_launch_us`); `spaced_launch_us` = the same launches with ~70 us of near-idle kernels between them (what the step
looks like to this kernel; an uninterrupted 22 GB stream runs ~5 % slower); `traffic` / `mfma_util` from the committed PMC
record profiles/r05_pmc_step.json, whose git head is echoed as `pmc_head` and which is withheld if csrc/moe.hip has changed
since.  `collectives` (N > 1): transport, per-launch GPU time of the fused all-reduce and of the logits all-gather, and
`per_transport`: the same norm + quant launch without a collective, with the xGMI all-reduce in its one-shot and its
two-shot form, and the all-gather, at bs 1 / 16 / 32.  A line whose `ranks_seen_by_library` differs from --gpus, or whose
xGMI error word is set, carries `invalid` (and no `value` in the first case).  `v2_lite`, `mixtral_8x7b_int8`, `llama3_8b`:
BASELINE configs 3 / 4 / 2 as extra objects with their own step_algorithmic_GB and roofline_frac (N = 1 only); `ep8_rank`: one
expert-parallel rank of R1 the same way.  `roofline_kernels`: the dominant kernel and the second expert GEMM
(W2) as a list, both timed as above.  `prefill`: a 2048-token prompt through one R1 rank shard -- ms per layer, and
the MFMA fraction (of the dense 2.5 PFLOP/s peak) of the MLA causal attention kernel and of the tiled fp8 GEMMs alone.

`box_calibration` / `value_normalised` (round 5): three probes of the box itself (device copy bandwidth, the per-launch time of a
16-workgroup dependent chain in a hipGraph, the W2 expert GEMM) and the step time mapped to a fixed reference box by a two-term model.

`graph_verified` (round 4): every timed hipGraph is replayed once more after its timed loop against the EAGER launches of the
same step on the same state (tokens, lengths, block table; the KV row at position L is rewritten with the same bytes) and
must equal them bit for bit -- per measured loop in `replay_vs_eager_after_timing`; `captures_checked_at_capture` /
`captures_rejected_and_repeated` echo chitu_amd.graphs.capture_verified's log (every capture is checked by one replay
before it is used).  Anything but equality voids the line (`invalid`, `value` null).  `cpu_baseline.reference_fields`
says which of its fields are static records (the reference's own CPU decode cannot be timed on the GPU box: no reference tree).

Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the bf16 attention kernels are priced against it)
EXTRA_BATCHES = tuple(int(b) for b in os.environ.get("CHITU_BENCH_EXTRA_BATCHES", "1,16").split(","))  # (profiling tools: one batch size alone)
FP8_MFMA_PEAK_TFLOPS = 5000.0  # dense fp8 peak (MX-scaled K = 128 form, measured 4.65 PF): what the fp8 GEMMs are priced against
PMC_FILE = "r06_pmc_step.json"  # tools/pmc_passes.sh + tools/pmc_report.py, stamped with git head + source digests


def _sha256(path):
    import hashlib

    with open(path, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()
SHARD = 8  # the metric's TP degree


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--bs", type=int, default=16)
    ap.add_argument("--ctx", type=int, default=1024)
    ap.add_argument("--layers", type=int, default=61, help="debug only: fewer layers => result flagged invalid")
    ap.add_argument("--router-std", type=float, default=None, help="debug only: synthetic router weight std (result flagged invalid)")
    ap.add_argument("--opt", type=str, default="", help="debug only: launch-variant overrides name=value,... (chitu_hip_debug_option; same results by construction, still reported in the line)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-graph-check", action="store_true", help="kernel-trace tools only: skip the replay-vs-eager check after each timed loop (its eager step would sit in the trace); the line is then flagged invalid")
    ap.add_argument("--no-bs1", action="store_true", help="skip the extra bs=1 and bs=32 measurements")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-calibration", action="store_true",
                    help="kernel-trace tools only: skip box_calibration (its probes' launches would sit in the trace)")
    ap.add_argument("--no-llama", action="store_true", help="skip the extra Llama-3-8B / DeepSeek-V2-Lite / Mixtral-int8 (BASELINE configs 2, 3, 4) measurements")
    return ap.parse_args()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: become `torch.distributed.run` with N ranks."""
    import socket

    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    if torch.cuda.device_count() < n:
        env["CHITU_BENCH_BACKEND"] = "gloo"  # every rank on cuda:0: functional check only
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execvpe(sys.executable, cmd, env)


def setup_dist(n):
    if n > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(n)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    info = {"backend": None, "ranks_seen": 1, "shared_device": False}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # functional check on a box with fewer GPUs than ranks (CHITU_BENCH_BACKEND=gloo, set by self_launch):
        # every rank on cuda:0, library collectives through gloo; such a run is flagged invalid
        if os.environ.get("CHITU_BENCH_BACKEND") == "gloo":
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
            info["shared_device"] = True
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        from chitu_amd import tensor_parallel as tp

        tp.init_tp(world, 1)
        info["backend"] = dist.get_backend()
        # the ranks the library itself sees: a sum of ones through the backend's own all-reduce
        ones = torch.ones(1, device="cuda")
        dist.all_reduce(ones)
        info["ranks_seen"] = int(ones.item())
    else:
        torch.cuda.set_device(0)
    assert world == n, f"--gpus {n} but WORLD_SIZE={world}"
    return rank, world, local, info


def enable_collectives(world, max_bs, vocab_local):
    """N > 1: the in-graph xGMI collectives unless CHITU_ALLREDUCE=rccl (or their self-test fails on any rank)."""
    if world == 1:
        return "none (one rank)"
    from chitu_amd import tensor_parallel as tp

    if os.environ.get("CHITU_ALLREDUCE", "xgmi") != "rccl" and tp.enable_xgmi(
            max_rows=max(max_bs, 32), max_dim=8192, gather_bytes=max(max_bs, 32) * vocab_local * 2):
        return "xgmi (hand-written push all-reduce / all-gather kernels inside the step's hipGraph)"
    return f"{dist.get_backend()} (library calls between hipGraph pieces)"


@torch.inference_mode()  # like the decode step: the capture touches state created under inference mode
def time_collectives(bs, dim, vocab_local, iters=200, batches=(1, 16, 32)):
    """GPU time per launch (`iters` launches captured in a hipGraph, events on the replay stream; every rank runs them) of
    the step's collectives ALONE, per transport, so that an N > 1 line can be read: the fused [all-reduce + residual add +
    RMSNorm + fp8 quant] launch on the xGMI kernels in its one-shot and two-shot form, the same launch WITHOUT the
    collective (chitu_hip_rmsnorm: what a one-rank step pays at that place), the logits all-gather, and the library's
    (RCCL's) all-reduce / all-gather of the same tensors, eager.  `allreduce_add_norm_quant_us` / `logits_all_gather_us`
    are at the bench's batch size on the form the step uses."""
    from chitu_amd import ops
    from chitu_amd import tensor_parallel as tp

    comm = tp.xgmi_comm()
    gen = torch.Generator(device="cuda").manual_seed(3)
    w = torch.ones(dim, device="cuda", dtype=torch.bfloat16)

    def graph_us(fn):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(iters):
                fn()
        g.replay()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        del g
        return round(e0.elapsed_time(e1) * 1e3 / iters, 2)

    def eager_us(fn, n=50):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return round(e0.elapsed_time(e1) * 1e3 / n, 2)

    out, table = {}, {}
    for b in sorted(set(batches) | {bs}):
        part = torch.randn(b, dim, device="cuda", generator=gen).to(torch.bfloat16)
        x = torch.randn(b, dim, device="cuda", generator=gen).to(torch.bfloat16)
        y = torch.randn(b, vocab_local, device="cuda", generator=gen).to(torch.bfloat16)
        row = {"message_KB_per_rank": round(b * dim * 2 / 1024, 1),
               "norm_quant_without_collective_us": graph_us(lambda: ops.rms_norm(x, w, 1e-6, out_bf16=False, quant="act", add=part))}
        if comm is not None:
            default = comm.two_shot_bytes
            comm.set_two_shot(1 << 60)
            row["xgmi_one_shot_fused_us"] = graph_us(lambda: comm.allreduce_rmsnorm(part, x, w, 1e-6, out_bf16=False, quant="act"))
            if (dim // 8) % comm.world == 0:
                comm.set_two_shot(0)
                row["xgmi_two_shot_fused_us"] = graph_us(lambda: comm.allreduce_rmsnorm(part, x, w, 1e-6, out_bf16=False, quant="act"))
            comm.set_two_shot(default)
            row["xgmi_form_in_step"] = "two-shot" if comm.uses_two_shot(b, dim) else "one-shot"
            if comm.gather_fits(b, vocab_local):
                row["xgmi_logits_all_gather_us"] = graph_us(lambda: comm.all_gather_last_dim(y, torch.float32))
        if dist.get_backend() == "nccl":  # the library's own collectives on the same tensors (eager: not captured here)
            t = part.clone()
            row["library_all_reduce_us"] = eager_us(lambda: dist.all_reduce(t))
            gathered = torch.empty(dist.get_world_size() * b, vocab_local, device="cuda", dtype=torch.bfloat16)
            row["library_all_gather_us"] = eager_us(lambda: dist.all_gather_into_tensor(gathered, y))
        table[f"bs{b}"] = row
    mine = table[f"bs{bs}"]
    if comm is not None:
        out["allreduce_add_norm_quant_us"] = mine["xgmi_two_shot_fused_us" if mine["xgmi_form_in_step"] == "two-shot" else "xgmi_one_shot_fused_us"]
        out["logits_all_gather_us"] = mine.get("xgmi_logits_all_gather_us")
    out["per_transport"] = table
    return out


class DeviceStateSampler:
    """rocm-smi clocks / power of the local GPU sampled from a host thread while a replica of the timed loop runs
    (boxes of the pool differ by up to 10 % on the latency-bound launches: the line carries what the device was doing)."""

    def __init__(self, device_index):
        import threading

        self.dev, self.rows, self.stop = str(device_index), [], threading.Event()
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import subprocess

        while not self.stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "-d", self.dev, "--showclocks", "--showpower", "--csv"], capture_output=True,
                                     text=True, timeout=5).stdout.strip().splitlines()
                if len(out) >= 2:
                    self.rows.append(dict(zip(out[0].split(","), out[-1].split(","))))
            except Exception:  # noqa: BLE001 -- diagnostics only
                pass
            self.stop.wait(0.2)

    def __enter__(self):
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self.stop.set()
        self.thread.join(timeout=10)
        return False

    def summary(self):
        import re

        def nums(key_part):
            vals = []
            for r in self.rows:
                for k, v in r.items():
                    if key_part in k.lower():
                        m = re.search(r"([0-9.]+)", v or "")
                        if m and "level" not in k.lower() and float(m.group(1)) > 10:  # clock speeds / watts, not level indices
                            vals.append(float(m.group(1)))
            return vals

        out = {"samples": len(self.rows)}
        for name, key in (("sclk_mhz", "sclk"), ("mclk_mhz", "mclk"), ("fclk_mhz", "fclk"), ("power_w", "power")):
            v = nums(key)
            if v:
                out[name] = {"min": min(v), "max": max(v)}
        return out


# ---- box calibration (round 5): the pool's boxes differ by up to 5 % on the same binary (rounds 2-4: 9.76 vs 10.28 ms at bs 16
# with identical sclk / mclk / fclk under load; the <= 32-workgroup launches and the W2 GEMM stretch on the slow ones).  Three
# probes of the box itself, run in this process right after the timed loops, and a two-term model that maps the step time to
# what the reference box of these constants would have measured.
CAL_REF = {"copy_GBs": 2550.0, "chain_us": 2.45,  # a "fast" box of rounds 4-5 (profiles/r05_box_calibration.txt)
           # tail_launch_probe on a fast box of round 6 (step 9.72 / 4.37 / 13.83 ms; profiles/r06_bench_default_fast_box.json)
           "tail_launch_us": {"chitu_hip_mla_q_proj": 7.34, "chitu_hip_gate_route_align": 8.32, "chitu_hip_rmsnorm": 4.46,
                              "chitu_hip_absorb_bmm_rope_fp8": 3.63, "chitu_hip_mla_merge_absorb_uv_quant_fp8_tm": 6.73}}
CAL_WEIGHTS = {"bandwidth": 0.55, "latency": 0.45}  # share of the bs-16 step's kernel time in HBM-bound launches (expert GEMMs,
# dense GEMMs, logits) / in the latency-bound tail (profiles/r04_step_breakdown_bs16_final.txt: 5.45 ms / 4.43 ms)


@torch.inference_mode()
def box_calibration(local_index, w2_launch_us=None):
    """{copy_GBs: 1 GB device-to-device copy (bytes copied per second; traffic is twice that), chain_us: one launch of a
    16-workgroup kernel inside a 400-launch dependent chain replayed as a hipGraph (the latency-bound tail's unit), w2_launch_us:
    the W2 expert GEMM at the step's own routing (roofline_kernels), static device facts, and `step_time_factor`: the factor
    by which this box's step is expected to be slower than the reference box's, 0.55 * ref_bw / bw + 0.45 * chain / ref_chain}."""
    from chitu_amd import ops

    dev = torch.device("cuda")
    src = torch.empty(1 << 30, dtype=torch.uint8, device=dev).fill_(1)
    dst = torch.empty_like(src)
    for _ in range(2):
        dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    copy_gbs = 10 * (1 << 30) / (e0.elapsed_time(e1) * 1e-3) / 1e9
    del src, dst
    # dependent chain: RMSNorm of 16 rows x 7168 (16 workgroups), each launch reading the previous one's output
    w = torch.ones(7168, dtype=torch.bfloat16, device=dev)
    x = torch.randn(16, 7168, device=dev).to(torch.bfloat16)
    for _ in range(3):
        y = ops.rms_norm(x, w, 1e-6)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    n_chain = 400
    with torch.cuda.graph(g):
        y = x
        for _ in range(n_chain):
            y = ops.rms_norm(y, w, 1e-6)
    g.replay()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    chain_us = e0.elapsed_time(e1) * 1e3 / (5 * n_chain)
    del g

    # dependent LOADS (round 5, second probe set): the slow boxes stretch exactly the launches that are chains of dependent
    # memory round trips (mla_q_proj 7.6 -> 9.9 us, route + align 8.9 -> 9.8) while the two probes above read the same there.
    # One-element gathers through a random cycle, 400 launches in a hipGraph: over a 512 MB table (every hop an HBM miss past
    # the 256 MB Infinity Cache) and over a 4 KB one (L2 hits); the difference is the memory round trip itself.
    def gather_chain(n_entries, hops=400):
        perm = torch.randperm(n_entries, device=dev)
        table = torch.empty(n_entries, dtype=torch.int64, device=dev)
        table[perm] = torch.roll(perm, -1)  # one cycle through all entries, in random order
        cur = perm[:1].clone()
        for _ in range(3):
            cur = table[cur]
        torch.cuda.synchronize()
        gg = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gg):
            c = cur
            for _ in range(hops):
                c = table[c]
        gg.replay()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            gg.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (5 * hops)
        del gg, table, perm
        return us

    hop_hbm_us = gather_chain(1 << 26)
    hop_l2_us = gather_chain(512)
    prop = torch.cuda.get_device_properties(0)
    out = {"copy_GBs": round(copy_gbs, 1), "chain_us": round(chain_us, 3), "w2_launch_us": w2_launch_us,
           "dependent_gather_us": {"hbm_table_512MB": round(hop_hbm_us, 3), "l2_table_4KB": round(hop_l2_us, 3),
                                   "memory_round_trip_ns": round((hop_hbm_us - hop_l2_us) * 1e3, 1)},
           "compute_units": prop.multi_processor_count, "device": prop.name, "hbm_GB": round(prop.total_memory / 1e9, 1),
           "reference_box": CAL_REF, "model_weights": CAL_WEIGHTS}
    out["step_time_factor"] = round(CAL_WEIGHTS["bandwidth"] * CAL_REF["copy_GBs"] / copy_gbs
                                    + CAL_WEIGHTS["latency"] * chain_us / CAL_REF["chain_us"], 4)
    try:
        import subprocess

        smi = subprocess.run(["rocm-smi", "-d", str(local_index), "--showperflevel", "--showmemuse", "--showclocks"],
                             capture_output=True, text=True, timeout=10).stdout
        keep = [ln.strip() for ln in smi.splitlines() if any(k in ln for k in ("Performance Level", "Memory", "sclk", "mclk", "fclk"))]
        out["rocm_smi_idle"] = keep[:12]
    except Exception as exc:  # noqa: BLE001 -- diagnostics only
        out["rocm_smi_idle"] = f"unavailable: {type(exc).__name__}"
    return out


TAIL_PROBE_ENTRIES = {  # C entry -> (kernel the step breakdown lists it as, which of a step's calls of that entry to keep)
    "chitu_hip_mla_q_proj": "mla_q_proj_kernel<1> (q_norm + wq_b GEMM + this step's KV rows)",
    "chitu_hip_gate_route_align": "gate_route_align_wg_kernel<32> (routing + moe_align in one workgroup)",
    "chitu_hip_rmsnorm": "rmsnorm_add_kernel (attn_norm / ffn_norm with the deferred top-k sum, residual add, act_quant)",
    "chitu_hip_absorb_bmm_rope_fp8": "absorb_bmm_kernel (W_UK absorb + RoPE of q_pe)",
    "chitu_hip_mla_merge_absorb_uv_quant_fp8_tm": "mla_merge_uv_quant_kernel (split merge + W_UV + act_quant)",
}


@torch.inference_mode()
def tail_launch_probe(model, cache, bs, ctx, iters=5):
    """the launches of the latency-bound tail that stretch on a slow box (mla_q_proj, route + align,
    the norm launches; round 5 saw 7.6 -> 9.9 us and 8.9 -> 9.8 us between boxes) timed IN THIS LINE, the way `roofline` times
    the expert GEMM: one eager decode step is recorded at the C ABI (chitu_amd._lib.call_log_ctypes: entry + argument objects),
    every recorded launch of an entry is issued again -- same pointers, one per layer, each on its own layer's weights -- back to
    back inside one hipGraph, HIP events around the replays; us per launch = replay / launches.  A chain of identical launches
    has no producer in front of it, so these are lower bounds of the in-step durations (the kernel trace has those), but they
    move with the box exactly as the in-step ones do."""
    from chitu_amd import _lib

    reqs = [f"tail{i}" for i in range(bs)]
    for r in reqs:
        cache.register_sequence(r, ctx)
    gen = torch.Generator(device="cuda").manual_seed(5)
    tokens = torch.randint(100, 1000, (bs,), device="cuda", generator=gen)
    cache.prepare_cache_decode(reqs)
    cache.prepare_block_table_for_decode(reqs)
    model.decode(tokens, use_graph=False)  # warm
    _lib.call_log_ctypes = []
    try:
        model.decode(tokens, use_graph=False)
        torch.cuda.synchronize()
        calls = _lib.call_log_ctypes
    finally:
        _lib.call_log_ctypes = None
    out = {}
    cdll = _lib.lib()
    for entry, label in TAIL_PROBE_ENTRIES.items():
        sel = [args for name, args in calls if name == entry]
        if len(sel) < 8:
            continue
        fn = getattr(cdll, entry)

        def issue_all():
            st = _lib.stream_ptr()
            for args in sel:
                rc = fn(*args[:-1], st)
                assert rc == 0, (entry, rc)

        try:
            issue_all()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                issue_all()
            g.replay()
            torch.cuda.synchronize()
            ev = []
            for _ in range(iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                g.replay()
                e1.record()
                ev.append((e0, e1))
            torch.cuda.synchronize()
            us = sorted(a.elapsed_time(b) * 1e3 / len(sel) for a, b in ev)
            out[entry] = {"kernel": label, "launches_in_graph": len(sel), "us_per_launch": round(us[len(us) // 2], 3),
                          "us_per_launch_min": round(us[0], 3)}
            del g
        except Exception as exc:  # noqa: BLE001 -- a diagnostic leg of the report
            out[entry] = {"kernel": label, "error": f"{type(exc).__name__}: {exc}"[:200]}
    for r in reqs:
        cache.finalize_cache_all_decode(r)
    return out


def static_device_facts(local_index):
    """Facts that do not change while the box lives, logged once: memory vendor, partition modes, voltage,
    xGMI error counters -- whatever rocm-smi on the box answers; missing tools or fields are recorded as such, never guessed."""
    import subprocess

    facts = {}
    for key, flags in {"unique_id": ["--showuniqueid"], "serial": ["--showserial"], "mem_vendor": ["--showmemvendor"], "partition": ["--showcomputepartition", "--showmemorypartition"],
                       "voltage": ["--showvoltage"], "xgmi_err": ["--showxgmierr"], "vbios": ["--showvbios"],
                       "power_cap": ["--showmaxpower"]}.items():
        try:
            res = subprocess.run(["rocm-smi", "-d", str(local_index)] + flags, capture_output=True, text=True, timeout=10)
            lines = [ln.strip() for ln in res.stdout.splitlines() if ln.strip().startswith("GPU[")]
            facts[key] = lines[:6] if lines else f"no answer (rc {res.returncode})"
        except Exception as exc:  # noqa: BLE001
            facts[key] = f"unavailable: {type(exc).__name__}"
    return facts


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()


def build_model(args_ns, rank):
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Args, DeepSeekV3Decoder, init_synthetic_

    margs = DeepSeekV3Args(shard_degree=SHARD, n_layers=args_ns.layers)
    max_seq = args_ns.ctx + args_ns.steps + args_ns.warmup + 256
    max_reqs = max(args_ns.bs, 1 if args_ns.no_bs1 else 32)
    cache = PagedKVCacheManager(0, margs.n_layers, num_hot_req=max_reqs, block_size=64, max_seq_len=max_seq,
                                device="cuda", kv_shape_per_sample=(margs.kv_lora_rank + margs.qk_rope_head_dim,),
                                dtype=torch.bfloat16)
    be = HipAttnBackend(local_n_heads=margs.n_heads // SHARD, max_seq_len=max_seq)
    model = DeepSeekV3Decoder(margs, cache, be, max_position_embeddings=max(max_seq, 4097), device="cuda")
    init_synthetic_(model, seed=1000 + rank, router_std=getattr(args_ns, 'router_std', None))
    # synthetic "prefilled" latent KV (prefill is out of scope)
    gen = torch.Generator(device="cuda").manual_seed(77 + rank)
    flat = cache.paged_kv_cache.view(-1)
    for i in range(0, flat.numel(), 1 << 26):
        n = min(1 << 26, flat.numel() - i)
        flat[i : i + n].copy_(torch.randn(n, device="cuda", dtype=torch.bfloat16, generator=gen) * 0.5)
    return margs, model, cache


def run_decode(model, cache, reqs, tokens, steps, world, use_graph, timed):
    """Exactly `steps` decode steps; returns (seconds (max over ranks) or None, final tokens)."""
    from chitu_amd import sampling

    if timed:
        barrier_sync(world)
        t0 = time.perf_counter()
    for _ in range(steps):
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        logits = model.decode(tokens, use_graph=use_graph)
        tokens = sampling.argmax(logits)  # greedy (executor.py:103-104): chitu_hip_sample, stays on device
        cache.finalize_cache_single_decode(reqs)
    if not timed:
        torch.cuda.synchronize()
        return None, tokens
    barrier_sync(world)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, tokens


NO_GRAPH_CHECK = False  # --no-graph-check (kernel-trace tools only: the check's eager step would sit in the trace)
GRAPH_CHECKS = {}  # tag -> one replay of the timed graph == one eager step on the same state (bitwise), per measured loop


def check_graph_against_eager(model, cache, reqs, tokens, use_graph, tag):
    """The graph that was just timed, once more, against the eager launches of the same step on the same state (same
    tokens, lengths, block table; the step's only side effect -- the KV row at position L -- is rewritten with the same
    bytes).  Bitwise, on every rank; a line whose graph differs from its eager step is void (main())."""
    if not use_graph or NO_GRAPH_CHECK:
        return
    cache.prepare_cache_decode(reqs)
    cache.prepare_block_table_for_decode(reqs)
    eager = model.decode(tokens, use_graph=False).clone()
    replay = model.decode(tokens, use_graph=use_graph)
    GRAPH_CHECKS[tag] = bool(torch.equal(eager, replay))


def measure(model, cache, bs, ctx, steps, warmup, world, use_graph, tag):
    reqs = [f"{tag}{i}" for i in range(bs)]
    for r in reqs:
        cache.register_sequence(r, ctx)
    gen = torch.Generator(device="cuda").manual_seed(5)
    tokens = torch.randint(100, 1000, (bs,), device="cuda", generator=gen)
    _, tokens = run_decode(model, cache, reqs, tokens, warmup, world, use_graph, timed=False)
    dt, tokens = run_decode(model, cache, reqs, tokens, steps, world, use_graph, timed=True)
    check_graph_against_eager(model, cache, reqs, tokens, use_graph, tag)
    for r in reqs:
        cache.finalize_cache_all_decode(r)
    return dt


def roofline_array(roof):
    """Both expert GEMMs (55 % of the bs-16 step) as a list, the dominant one first: `roofline` stays the dominant kernel's
    object (the bench contract), this is the same plus the second kernel."""
    if not roof:
        return None
    first = {k: roof[k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_us", "algorithmic_bytes_per_launch")
             if k in roof}
    return [first] + ([roof["second_kernel"]] if roof.get("second_kernel") else [])


def graph_report(world=1):
    """What stands behind every graph this run timed: the replay checks above, and the capture-time checks of
    chitu_amd.graphs.capture_verified (a capture whose first replay differs from the eager step is rejected and repeated;
    any such event is listed).  Both verdicts are reduced (MIN) over the ranks: `all_equal_eager` is true only if every
    rank's replays equalled its eager steps, `captures_clean_on_every_rank` only if no rank repeated a capture."""
    from chitu_amd import graphs

    ok_local = all(GRAPH_CHECKS.values()) if GRAPH_CHECKS else None
    clean_local = not graphs.unverified_or_retried()
    ok_all, clean_all = ok_local, clean_local
    if world > 1:
        v = torch.tensor([1 if ok_local else 0, 1 if clean_local else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        ok_all = bool(v[0].item()) if ok_local is not None else None
        clean_all = bool(v[1].item())
    return {"all_equal_eager": ok_all, "captures_clean_on_every_rank": clean_all,
            "replay_vs_eager_after_timing": dict(GRAPH_CHECKS), "captures_checked_at_capture": len(graphs.capture_log),
            "captures_rejected_and_repeated": graphs.unverified_or_retried()}


def capture_step_routing(model, cache, bs, ctx):
    """One eager decode step on fresh sequences with a hook on every router: returns the
    [bs, topk+1] expert ids each MoE layer actually routed (shared expert = last column)."""
    from chitu_amd.deepseek_v3 import GateDeepSeekV3

    rec = []
    hooks = [m.register_forward_hook(lambda mod, i, o: rec.append(o[1].clone()))
             for m in model.modules() if isinstance(m, GateDeepSeekV3)]
    reqs = [f"route{i}" for i in range(bs)]
    for r in reqs:
        cache.register_sequence(r, ctx)
    gen = torch.Generator(device="cuda").manual_seed(5)
    tokens = torch.randint(100, 1000, (bs,), device="cuda", generator=gen)
    for _ in range(2):  # second step: tokens are the model's own greedy picks, as in the timed run
        rec.clear()
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        tokens = model.decode(tokens, use_graph=False).argmax(dim=-1)
        cache.finalize_cache_single_decode(reqs)
    torch.cuda.synchronize()
    for h in hooks:
        h.remove()
    for r in reqs:
        cache.finalize_cache_all_decode(r)
    return rec


def algorithmic_bytes_per_step(margs, bs, ctx, distinct):
    """every weight byte counted once per step, per GPU, TP = SHARD."""
    t = SHARD
    d = margs.dim
    attn = (margs.q_lora_rank + margs.kv_lora_rank + margs.qk_rope_head_dim) * d \
        + (margs.n_heads * 192 // t) * margs.q_lora_rank + (margs.n_heads * 256 // t) * margs.kv_lora_rank \
        + d * (margs.n_heads * 128 // t)
    dense = 3 * margs.inter_dim * d // t
    moe = margs.n_routed_experts * d * 2 + (1 + distinct) * 3 * margs.moe_inter_dim * d // t
    n_moe = margs.n_layers - margs.n_dense_layers
    head = (margs.vocab_size // t) * d * 2
    kv = margs.n_layers * bs * ctx * 576 * 2
    return margs.n_layers * attn + margs.n_dense_layers * dense + n_moe * moe + head + kv


@torch.inference_mode()  # like the decode step: the capture touches state created under inference mode
def roofline_dominant_kernel(model, routing, margs, bs, iters=3):
    """Time the dominant kernel -- the routed-expert GEMM1 with SiLU-and-mul in its epilogue
    (chitu_hip_moe_gemm1_silu_fp8 -> moe_gemm1_silu_kernel: ~2/3 of all bytes at bs=16) -- live with
    HIP events on the launch stream: one launch per MoE layer with that layer's own weights (HBM-cold) and the
    expert ids that layer really routed in a decode step (capture_step_routing), captured into one hipGraph and
    replayed, i.e. the same launches in the same form as the timed step's graph."""
    from chitu_amd import _lib, fused_moe
    from chitu_amd._lib import f32, i32, i64, ptr, stream_ptr

    lib = _lib.lib()
    moe_layers = [l.ffn for l in model.layers if l.is_moe]
    if not moe_layers or not routing:
        return None
    assert len(routing) == len(moe_layers)
    E, K = margs.n_routed_experts + margs.n_shared_experts, margs.dim
    N = moe_layers[0].w1w3_weight.shape[1]
    topk = routing[0].shape[1]
    numel = bs * topk
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.randn(bs, K, device="cuda", dtype=torch.bfloat16, generator=gen)
    xq, xs = fused_moe.per_token_group_quant_fp8(x, 128)
    out = torch.empty(numel, N // 2, dtype=torch.bfloat16, device="cuda")
    plans = []
    for ids in routing:
        sorted_ids, expert_ids, npost = fused_moe.moe_align_block_size(ids.contiguous(), 16, E)
        plans.append((sorted_ids, expert_ids, npost, int(ids.unique().numel())))
    def launch(m, plan):
        sorted_ids, expert_ids, npost, _ = plan
        rc = lib.chitu_hip_moe_gemm1_silu_fp8(ptr(xq), ptr(xs), ptr(m.w1w3_weight), ptr(m.w1w3_scale), ptr(sorted_ids),
                                              ptr(expert_ids), ptr(npost), ptr(out), i64(numel), i32(topk), i64(N // 2),
                                              i64(K), i64(min(expert_ids.numel(), numel)), stream_ptr())
        assert rc == 0

    for m, pl in list(zip(moe_layers, plans))[:2]:
        launch(m, pl)
    torch.cuda.synchronize()
    # (1) the launches the way the step issues them: one hipGraph holding every MoE layer's launch back to back (each
    # with its own HBM-cold weights), an event pair around each replay; launch duration = replay time / launches.
    # This is what rocprofv3's per-kernel average of the timed step measures (profiles/r02_step_breakdown_bs16_final.txt).
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for m, pl in zip(moe_layers, plans):
            launch(m, pl)
    graph.replay()
    torch.cuda.synchronize()
    reps = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        reps.append((e0, e1))
    torch.cuda.synchronize()
    per_launch = sorted(a.elapsed_time(b) / len(plans) for a, b in reps)
    avg_ms = sum(per_launch) / len(per_launch)
    # (1b) the same launches with a near-idle gap after each (a chain of 32 one-workgroup elementwise kernels, ~70 us: what
    # the rest of a layer lasts in the step), minus a graph of the gaps alone: separates "58 launches streaming 22 GB without a pause"
    # (HBM refresh / power management under an uninterrupted stream) from the kernel's own duration.  In the step the
    # kernel trace shows the in-step figure; this leg reproduces it without a profiler.
    spaced_ms = None
    try:
        tick = torch.zeros(64, device="cuda")

        def gap():
            for _ in range(32):
                tick.add_(1.0)

        gap()
        torch.cuda.synchronize()
        g_gap, g_spaced = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_gap):
            for _ in plans:
                gap()
        with torch.cuda.graph(g_spaced):
            for m, pl in zip(moe_layers, plans):
                launch(m, pl)
                gap()
        both = []
        for g in (g_gap, g_spaced, g_gap, g_spaced, g_gap, g_spaced):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            both.append((e0, e1))
        torch.cuda.synchronize()
        t = [a.elapsed_time(b) for a, b in both]
        gap_ms, with_ms = min(t[0::2]), min(t[1::2])
        spaced_ms = (with_ms - gap_ms) / len(plans)
        gap_us = gap_ms / len(plans) * 1e3
    except Exception as e:
        print(f"[bench] spaced-launch leg skipped: {e!r}", file=sys.stderr)
        gap_us = None
    # (2) for comparison, an event pair around every single eager launch: the interval then also holds the two
    # event packets and the launch itself (~4 us), which the step's graph does not pay
    times = []
    for m, pl in zip(moe_layers, plans):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch(m, pl)
        e1.record()
        times.append((e0, e1))
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in times)
    pair_avg_ms = sum(ms) / len(ms)
    # algorithmic bytes of the average launch: each hit expert's W1 slice once + its scales +
    # activations + output
    distinct = sum(pl[3] for pl in plans) / len(plans)
    w_bytes = distinct * N * K
    s_bytes = distinct * ((N + 127) // 128) * (K // 128) * 4
    a_bytes = bs * K + bs * (K // 128) * 4 + numel * (N // 2) * 2
    alg = w_bytes + s_bytes + a_bytes
    achieved = alg / (avg_ms * 1e-3) / 1e9
    # HBM traffic and MFMA utilisation of this kernel from the committed counter passes (tools/pmc_passes.sh ->
    # tools/pmc_report.py -> profiles/r02_pmc_step.json: FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE and
    # SQ_VALU_MFMA_BUSY_CYCLES in their own passes), taken on an eager step of this same synthetic model at bs 16
    traffic = mfma_util = pmc_head = pmc_note = None
    try:
        # the counters are a committed record (PMC passes cost minutes), so they carry the identity of the code they were
        # taken on: the git head and a digest of the kernel's source.  A digest that differs from the tree's voids them.
        doc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
        pmc_head = doc.get("git_head")
        k = next(v for n, v in doc["kernels"].items() if "moe_gemm1_silu_kernel" in n)
        if doc.get("source_sha256", {}).get("moe.hip") == _sha256(os.path.join(ROOT, "chitu_amd", "csrc", "moe.hip")):
            traffic = int((k["hbm_read_MB"] + k.get("hbm_write_MB_uncalibrated", 0.0)) * 1e6)
            mfma_util = k.get("mfma_util")
        else:
            pmc_note = f"profiles/{PMC_FILE} was taken on another version of csrc/moe.hip: traffic / mfma_util withheld"
    except Exception:
        pass
    # ---- the second expert GEMM the same way (one hipGraph of every MoE layer's launch back to back, each on its own
    # HBM-cold weights and that layer's routing; HIP events around the replays): W2 slices [dim, I] of the hit experts
    gemm2 = None
    try:
        I = N // 2
        wts = torch.rand(bs, topk, device="cuda", generator=gen).to(torch.bfloat16)
        out2 = torch.empty(numel, K, dtype=torch.bfloat16, device="cuda")

        def launch2(m, plan):
            sorted_ids, expert_ids, npost, _ = plan
            rc = lib.chitu_hip_moe_gemm2_quant_fp8(ptr(out), ptr(m.w2_weight), ptr(m.w2_scale), ptr(sorted_ids), ptr(expert_ids),
                                                   ptr(npost), ptr(wts), i32(0), i32(1), ptr(out2), i64(numel), i64(K), i64(I),
                                                   i64(min(expert_ids.numel(), numel)), f32(1e-10), stream_ptr())
            assert rc == 0

        launch2(moe_layers[0], plans[0])
        torch.cuda.synchronize()
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2):
            for m, pl in zip(moe_layers, plans):
                launch2(m, pl)
        g2.replay()
        torch.cuda.synchronize()
        r2 = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g2.replay()
            e1.record()
            r2.append((e0, e1))
        torch.cuda.synchronize()
        avg2_ms = sum(a.elapsed_time(b) for a, b in r2) / len(r2) / len(plans)
        alg2 = distinct * K * I + distinct * (K // 128) * ((I + 127) // 128) * 4 + numel * I * 2 + numel * K * 2 + numel * 2
        gemm2 = {"kernel": "moe_gemm2_q_kernel (routed experts W2, per-group re-quantisation of h in the prologue, routed weight in the epilogue)",
                 "bound": "hbm", "achieved": round(alg2 / (avg2_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(alg2 / (avg2_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "avg_launch_us": round(avg2_ms * 1e3, 2),
                 "algorithmic_bytes_per_launch": int(alg2),
                 "note": "of which written: the un-summed top-k slots [bs * (topk + shared), dim] bf16 (the sum rides in the next "
                         "norm launch); timing as for the first kernel"}
    except Exception as e:  # noqa: BLE001 -- a leg of the report, not of the measurement
        print(f"[bench] second expert GEMM leg skipped: {e!r}", file=sys.stderr)
    return {
        "second_kernel": gemm2,
        "kernel": "moe_gemm1_silu_kernel (routed experts W1, fp8 block-scaled grouped GEMM + SiLU-and-mul epilogue)",
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "mfma_util": mfma_util, "pmc_head": pmc_head,
        "pmc_note": pmc_note,
        "achievable_note": "the chip's measured copy bandwidth is 6.29 TB/s = 0.79 of the 8 TB/s specification "
                           "(MI355X_MICROARCH.md); a pure cold READ of this launch's own size (395 MB in one pass, register ring, no "
                           "arithmetic: tools/probe_cold_stream.hip, profiles/r06_probe_cold_stream.txt) tops at 6.6 TB/s = 0.82, and "
                           "only a 1.9 GB launch amortises ramp and tail up to 7.0 = 0.88: 0.82, not 1.0, is this kernel's ceiling",
        "traffic_source": "rocprofv3 --pmc FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate passes, per launch of this kernel "
                          f"in an eager bs-16 step of the same model (profiles/{PMC_FILE}); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES "
                          "/ (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs), its own pass",
        "avg_launch_us": round(avg_ms * 1e3, 2),
        "timing": f"HIP events around {iters} replays of one hipGraph holding the {len(plans)} launches back to back (the step's own form); "
                  "avg_launch_us = replay time / launches",
        "spaced_launch_us": None if spaced_ms is None else round(spaced_ms * 1e3, 2),
        "spaced_note": None if spaced_ms is None else f"the same graph with a {gap_us:.0f} us idle gap after every launch, minus a graph of the gaps "
                       "alone: the launch as the step sees it (other kernels between two of them) rather than 58 of them streaming without a pause",
        "event_pair_avg_launch_us": round(pair_avg_ms * 1e3, 2), "event_pair_median_launch_us": round(ms[len(ms) // 2] * 1e3, 2),
        "algorithmic_bytes_per_launch": int(alg), "distinct_experts": round(distinct, 2),
        "distinct_experts_min_max": [min(pl[3] for pl in plans), max(pl[3] for pl in plans)],
        "routing": "expert ids of a real decode step of this model, per layer (shared expert included)",
        "launches_timed": iters * len(plans),
    }


def cpu_baseline(margs, bs, ctx):
    """The oracle (CPU restatement of the reference's decode path) timed on this host's cores, on a
    bounded sample: ONE MoE decoder layer at the full per-rank shapes and the full batch/context.
    Scaled by the layer count it gives the reference-CPU-path rate for the same per-GPU workload."""
    import torch.nn.functional as F  # noqa: F401

    from oracle import deepseek as ods

    # torch's intra-op pool collapses on the oracle's many small ops when given hundreds of
    # threads (measured: 158 s/layer with 256 threads vs <10 s with 16), so cap it; `cores` reports
    # the threads actually used.
    cores = min(os.cpu_count() or 1, 16)
    torch.set_num_threads(cores)
    H = margs.n_heads // SHARD
    I = margs.moe_inter_dim // SHARD
    # all 256 routed experts + the shared one are materialised (1.4 GB of host weights): 32 distinct random experts
    # tiled 8x + 1 -- generating 257 would take longer than using them, and the oracle's time does not depend on the
    # values; every token runs the real top-8-of-256 routing (8 groups / 4 limited) + the shared expert
    E, E_GEN = margs.n_routed_experts, 32
    d = margs.dim
    g = torch.Generator().manual_seed(0)
    fp8 = torch.float8_e4m3fn

    def w(*shape):
        return (torch.randn(*shape, generator=g, dtype=torch.bfloat16) * 0.5).to(fp8)

    def s(*shape):
        return torch.rand(*shape, generator=g) * 0.02 + 0.01

    pre = "layers.3."
    p = {
        pre + "attn_norm.weight": torch.ones(d, dtype=torch.bfloat16),
        pre + "ffn_norm.weight": torch.ones(d, dtype=torch.bfloat16),
        pre + "attn.wqkv_a.weight": w(2112, d), pre + "attn.wqkv_a.scale": s(17, d // 128),
        pre + "attn.q_norm.weight": torch.ones(1536, dtype=torch.bfloat16),
        pre + "attn.wq_b.weight": w(H * 192, 1536), pre + "attn.wq_b.scale": s(H * 192 // 128, 12),
        pre + "attn.kv_norm.weight": torch.ones(512, dtype=torch.bfloat16),
        pre + "attn.wkv_b.weight": w(H * 256, 512), pre + "attn.wkv_b.scale": s(H * 2, 4),
        pre + "attn.wo.weight": w(d, H * 128), pre + "attn.wo.scale": s(d // 128, H),
        pre + "ffn.gate.weight": (torch.randn(E, d, generator=g) * d ** -0.5).to(torch.bfloat16),
        pre + "ffn.gate.bias": (torch.randn(E, generator=g) * 0.01).to(torch.bfloat16),
    }

    def tiled(t):  # [E_GEN, ...] -> [E + 1, ...] real memory (fp8 has no repeat: go through bytes)
        reps = (E + 1 + E_GEN - 1) // E_GEN
        raw = t.view(torch.uint8) if t.element_size() == 1 else t
        out = raw.repeat(reps, *([1] * (t.dim() - 1)))[: E + 1].contiguous()
        return out.view(t.dtype) if t.element_size() == 1 else out

    p[pre + "ffn.w1w3_weight"], p[pre + "ffn.w1w3_scale"] = tiled(w(E_GEN, 2 * I, d)), tiled(s(E_GEN, 2 * I // 128, d // 128))
    p[pre + "ffn.w2_weight"], p[pre + "ffn.w2_scale"] = tiled(w(E_GEN, d, I)), tiled(s(E_GEN, d // 128, I // 128))
    from chitu_amd.deepseek_v3 import compute_softmax_scale

    cfg = dict(H=H, C=512, R=64, NOPE=128, V=128, QL=1536, eps=1e-6, scale=compute_softmax_scale(margs),
               n_groups=8, topk_groups=4, topk=8, score_func="sigmoid", route_scale=2.5, n_routed=E)
    ods.block(p, 3, torch.randn(1, d, generator=g).to(torch.bfloat16), torch.randn(1, 32, generator=g),
              torch.randn(1, 32, generator=g), (torch.randn(2, 64, 576, generator=g) * 0.5).to(torch.bfloat16),
              torch.zeros(1, 2, dtype=torch.int32), torch.full((1,), 10, dtype=torch.int32), cfg, True)  # warm-up
    pages_per = (ctx + 64) // 64 + 1
    cache = (torch.randn(bs * pages_per, 64, 576, generator=g) * 0.5).to(torch.bfloat16)
    table = torch.arange(bs * pages_per, dtype=torch.int32).view(bs, pages_per)
    lens = torch.full((bs,), ctx, dtype=torch.int32)
    cos = torch.randn(bs, 32, generator=g)
    sin = torch.randn(bs, 32, generator=g)
    x = torch.randn(bs, d, generator=g).to(torch.bfloat16)
    t0 = time.perf_counter()
    reps = 0
    while True:
        ods.block(p, 3, x, cos, sin, cache, table, lens, cfg, True)
        reps += 1
        if time.perf_counter() - t0 > 8.0 or reps >= 3:
            break
    per_layer = (time.perf_counter() - t0) / reps
    step_s = per_layer * margs.n_layers
    out = {
        "value": round((1.0 / SHARD) * bs / step_s, 4), "unit": "tok/s (same normalisation as value, 1 of 8 shards)",
        "cores": cores, "kind": "port",
        "sample": f"{reps} x one MoE decoder layer (per-rank R1 shapes, all 257 experts, bs={bs}, ctx={ctx}) on the CPU oracle, "
                  f"{per_layer * 1e3:.0f} ms/layer, scaled x{margs.n_layers} layers",
        "ms_per_layer": round(per_layer * 1e3, 1),
    }
    # the REFERENCE'S OWN decode timed on CPU (tools/time_reference_cpu.py; /root/reference exists in the build container
    # only, so it cannot be re-timed here): TransformerDeepSeekV3 at the same per-rank shapes -- its FP8 linears are Triton
    # kernels, which on a CPU run under TRITON_INTERPRET=1 -- and BASELINE config 1 (Llama-2-7B bf16, bs 1, 64 tokens)
    try:
        ref = json.load(open(os.path.join(ROOT, "profiles", "r02_ref_cpu_decode.json")))
        r = ref.get(f"deepseek_r1_tp8_rank_bs{bs}") or ref.get("deepseek_r1_tp8_rank_bs16")
        if r:
            out["reference_ms_per_layer"] = round(r["s_per_layer"] * 1e3, 1)
            out["reference_tok_s"] = round((1.0 / SHARD) * r["bs"] / (r["s_per_layer"] * margs.n_layers), 7)
            out["reference_sample"] = (f"{r['what']}; bs {r['bs']}, {r['threads']} threads on {ref['host']['cores']} cores, "
                                       f"{ref['host']['where']}: profiles/r02_ref_cpu_decode.json")
        if "llama2_7b_bf16_bs1" in ref:
            out["reference_config1_llama2_7b"] = ref["llama2_7b_bf16_bs1"]
        # every `reference_*` field above is STATIC: read from a committed record, not timed in this run
        out["reference_fields"] = {"static": True, "file": "profiles/r02_ref_cpu_decode.json",
                                   "cores": ref.get("host", {}).get("cores"), "where": ref.get("host", {}).get("where"),
                                   "why": "the reference tree exists in the build container only; the GPU box cannot import it, "
                                          "so its CPU decode cannot be timed beside the port here"}
    except Exception:  # noqa: BLE001 -- the live port above is the baseline; the reference numbers are a committed record
        pass
    return out


def llama3_8b_extra(steps, warmup, ctx):
    """BASELINE config 2 as an extra line: Llama-3-8B bf16, TP=1 (the whole model on this GPU), paged KV
    (page 256), hipGraph decode, synthetic weights; bs=1 and bs=16.  Algorithmic bytes per step = every
    weight once (16.06 GB) + the KV read."""
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.llama import LlamaArgs, LlamaDecoder, init_synthetic_

    args = LlamaArgs()
    max_seq = max(ctx + steps + warmup + 512, 2048 + 256)
    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=16, block_size=256, max_seq_len=max_seq, device="cuda",
                                n_local_kv_heads=args.n_kv_heads, head_dim=args.head_dim, dtype=torch.bfloat16)
    model = LlamaDecoder(args, cache, HipAttnBackend(local_n_heads=args.n_heads, max_seq_len=max_seq),
                         max_position_embeddings=max_seq, device="cuda")
    init_synthetic_(model, seed=3)
    cache.paged_k_cache.normal_(0, 0.5)
    cache.paged_v_cache.normal_(0, 0.5)
    w_bytes = sum(p.numel() * 2 for n, p in model.named_parameters() if n != "embed_weight")
    out = {"model": "Llama-3-8B bf16 TP=1, paged KV (page 256), hipGraph, synthetic weights", "weight_GB": round(w_bytes / 1e9, 3)}
    for bs in EXTRA_BATCHES:
        dt = measure(model, cache, bs, ctx, steps, warmup, 1, True, f"l{bs}_")
        kv = args.n_layers * bs * ctx * args.n_kv_heads * args.head_dim * 2 * 2
        out[f"bs{bs}"] = {"ms_per_step": round(dt / steps * 1e3, 4), "tok_s": round(bs * steps / dt, 1),
                          "hbm_GBs": round((w_bytes + kv) / (dt / steps) / 1e9, 1),
                          "roofline_frac": round((w_bytes + kv) / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4)}
    # config 2's prefill (SURVEY 8f.1 for the GQA family): one 2048-token prompt through the whole model (eager launches), and
    # the causal GQA attention kernel alone against the dense bf16 MFMA peak, beside the round-2 decode composition it replaced
    T = 2048
    gp = torch.Generator().manual_seed(0)
    prompt = torch.randint(100, 1000, (T,), generator=gp).tolist()
    times = []
    for rep in range(3):
        rid = f"lp{rep}"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.prefill([prompt], [rid])
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        cache.finalize_cache_all_decode(rid)
    be = model.layers[0].attn.attn_backend
    gd = torch.Generator(device="cuda").manual_seed(5)
    q = (torch.randn(T, args.n_heads, args.head_dim, device="cuda", generator=gd) * 0.5).to(torch.bfloat16)
    k = torch.randn(T, args.n_kv_heads, args.head_dim, device="cuda", generator=gd).to(torch.bfloat16)
    v = torch.randn(T, args.n_kv_heads, args.head_dim, device="cuda", generator=gd).to(torch.bfloat16)
    cu = torch.tensor([0, T], dtype=torch.int32, device="cuda")
    flop = 2.0 * args.n_heads * (T * (T + 1) / 2) * 2 * args.head_dim
    att, prev = {}, os.environ.get("CHITU_GQA_PREFILL")
    for mode, n in (("flash", 20), ("compose", 2)):
        os.environ["CHITU_GQA_PREFILL"] = mode
        for _ in range(2):
            be.attn_varlen_func(q, k, v, cu, cu, T, T, causal=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            be.attn_varlen_func(q, k, v, cu, cu, T, T, causal=True)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        att[mode] = {"us": round(us, 1), "TFLOPs": round(flop / us * 1e-6, 1), "frac": round(flop / us * 1e-6 / MFMA_PEAK_TFLOPS, 4)}
    if prev is None:
        os.environ.pop("CHITU_GQA_PREFILL", None)
    else:
        os.environ["CHITU_GQA_PREFILL"] = prev
    out["prefill_2048"] = {"ms": round(min(times[1:]) * 1e3, 3), "prompt_tok_s": round(T / min(times[1:]), 1),
                           "attention": {"kernel": "chitu::gqa_prefill_flash_kernel<4>", "bound": "mfma", "peak_TFLOPs": MFMA_PEAK_TFLOPS,
                                         **att["flash"], "decode_composition": att["compose"]}}
    del model, cache
    torch.cuda.empty_cache()
    return out


def v2_lite_extra(steps, warmup, ctx):
    """BASELINE config 3 as an extra line: DeepSeek-V2-Lite shapes (dim 2048, 27 layers, 16 heads, q_lora 0,
    64 routed + 2 shared experts, top-6, softmax router), FP8 block-scaled synthetic weights, TP=1, absorb."""
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Args, DeepSeekV3Decoder, init_synthetic_

    # inter_dim of the one dense layer: 10944 in the checkpoint, not a multiple of the 128 FP8 block (the
    # reference's act_quant asserts on it, ops.py:345-348); rounded up to 11008 for this synthetic run
    args = DeepSeekV3Args(vocab_size=102400, dim=2048, inter_dim=11008, moe_inter_dim=1408, n_layers=27, n_dense_layers=1,
                          n_heads=16, n_routed_experts=64, n_shared_experts=2, n_activated_experts=6, n_expert_groups=1,
                          n_limited_groups=1, route_scale=1.0, score_func="softmax", q_lora_rank=0, gate_bias=False,
                          shard_degree=1)
    max_seq = ctx + steps + warmup + 256
    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=16, block_size=64, max_seq_len=max_seq, device="cuda",
                                kv_shape_per_sample=(576,), dtype=torch.bfloat16)
    model = DeepSeekV3Decoder(args, cache, HipAttnBackend(local_n_heads=16, max_seq_len=max_seq),
                              max_position_embeddings=max(max_seq, 4097), device="cuda")
    init_synthetic_(model, seed=5)
    cache.paged_kv_cache.normal_(0, 0.5)
    out = {"model": "DeepSeek-V2-Lite shapes, FP8 block-scaled weights, TP=1, MLA absorb, hipGraph, synthetic weights"}
    H, d = args.n_heads, args.dim
    attn_w = (H * 192 + 576) * d + H * 256 * args.kv_lora_rank + d * H * 128  # wq|wkv_a, wkv_b, wo (fp8: 1 B / weight)
    head_w = args.vocab_size * d * 2
    n_moe = args.n_layers - args.n_dense_layers
    for bs in EXTRA_BATCHES:
        dt = measure(model, cache, bs, ctx, steps, warmup, 1, True, f"v{bs}_")
        # experts a step streams: the routed ones its own router picks (measured on an eager step) + both shared ones
        routing = capture_step_routing(model, cache, bs, ctx)
        distinct = sum(int(r[:, : args.n_activated_experts].unique().numel()) for r in routing) / max(1, len(routing))
        moe_w = args.n_routed_experts * d * 2 + (distinct + args.n_shared_experts) * 3 * args.moe_inter_dim * d
        alg = args.n_layers * attn_w + args.n_dense_layers * 3 * args.inter_dim * d + n_moe * moe_w + head_w \
            + args.n_layers * bs * ctx * 576 * 2
        out[f"bs{bs}"] = {"ms_per_step": round(dt / steps * 1e3, 4), "tok_s": round(bs * steps / dt, 1),
                          "distinct_routed_experts": round(distinct, 2), "step_algorithmic_GB": round(alg / 1e9, 3),
                          "roofline_frac": round(alg / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4)}
    del model, cache
    torch.cuda.empty_cache()
    return out


def prefill_extra(prompt_tokens=2048, n_moe_layers=6):
    """SURVEY 8f.1 as an extra object: ONE prompt of `prompt_tokens` tokens through `DeepSeekV3Decoder.prefill` of an R1 TP=8
    rank shard (3 dense + `n_moe_layers` MoE layers at the true per-layer shapes; eager launches, as the reference's prefill is),
    per-layer time from HIP events around every layer; then the two MFMA-bound kernel families alone, the bf16 attention against
    the dense 2.5 PFLOP/s bf16 peak and the fp8 GEMMs against the dense 5 PFLOP/s fp8 peak (round 6: they run on the K = 128 form): the MLA causal attention (chitu_hip_mla_prefill_flash; FLOPs = 2 x 16 heads x
    T(T+1)/2 pairs x (576 + 512) MACs) and the tiled block-scaled fp8 GEMMs of one layer at M = T (2 M N K each).  The
    bit-exact attention kernel (CHITU_MLA_PREFILL=exact) is timed beside it."""
    from chitu_amd import ops
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Args, DeepSeekV3Decoder, init_synthetic_

    T = prompt_tokens
    args = DeepSeekV3Args(shard_degree=SHARD, n_layers=3 + n_moe_layers)
    max_seq = T + 64
    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=2, block_size=64, max_seq_len=max_seq, device="cuda",
                                kv_shape_per_sample=(576,), dtype=torch.bfloat16)
    be = HipAttnBackend(local_n_heads=args.n_heads // SHARD, max_seq_len=max_seq)
    model = DeepSeekV3Decoder(args, cache, be, max_position_embeddings=max(max_seq, 4097), device="cuda")
    init_synthetic_(model, seed=11)
    g = torch.Generator().manual_seed(0)
    prompt = torch.randint(100, 1000, (T,), generator=g).tolist()
    marks = []

    def pre(_m, _i):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        marks.append(e)

    hooks = [layer.register_forward_pre_hook(pre) for layer in model.layers]
    best_total, best_layers = None, None
    for rep in range(4):
        marks.clear()
        rid = f"pf{rep}"
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.prefill([prompt], [rid])
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        torch.cuda.synchronize()
        total = time.perf_counter() - t0
        cache.finalize_cache_all_decode(rid)
        ev = marks + [end]  # the last interval also holds the final norm + head GEMM of ONE row: negligible
        per = [ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)]
        if rep and (best_total is None or total < best_total):
            best_total, best_layers = total, per
    for h in hooks:
        h.remove()
    dense_ms = sum(best_layers[:3]) / 3
    moe_ms = sorted(best_layers[3:])[len(best_layers[3:]) // 2]  # median MoE layer
    out = {"workload": f"DeepSeek-R1 FP8 TP=8 rank shard, one {T}-token prompt, eager prefill, 3 dense + {n_moe_layers} MoE layers",
           "ms_per_moe_layer": round(moe_ms, 4), "ms_per_dense_layer": round(dense_ms, 4),
           "rank_prompt_tok_s_at_61_layers": round(T / ((3 * dense_ms + 58 * moe_ms) * 1e-3), 1)}
    del model, cache
    torch.cuda.empty_cache()

    def time_us(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n

    gd = torch.Generator(device="cuda").manual_seed(5)
    q = (torch.randn(T, 16, 576, device="cuda", generator=gd) * 0.3).to(torch.bfloat16)
    kv = torch.randn(T, 1, 576, device="cuda", generator=gd).to(torch.bfloat16)
    cu = torch.tensor([0, T], dtype=torch.int32, device="cuda")
    flop = 2.0 * 16 * (T * (T + 1) / 2) * (576 + 512)
    prev = os.environ.get("CHITU_MLA_PREFILL")
    att = {}
    for mode in ("flash", "exact"):
        os.environ["CHITU_MLA_PREFILL"] = mode
        us = time_us(lambda: be.attn_varlen_func(q, kv, kv[..., :512], cu, cu, T, T, causal=True, softmax_scale=0.1352))
        att[mode] = {"us": round(us, 1), "TFLOPs": round(flop / us * 1e-6, 1), "frac": round(flop / us * 1e-6 / MFMA_PEAK_TFLOPS, 4)}
    if prev is None:
        os.environ.pop("CHITU_MLA_PREFILL", None)
    else:
        os.environ["CHITU_MLA_PREFILL"] = prev
    out["attention"] = {"kernel": "chitu::mla_prefill_flash_pipe_kernel", "bound": "mfma", "peak_TFLOPs": MFMA_PEAK_TFLOPS,
                        "GFLOP": round(flop * 1e-9, 2), **att["flash"], "exact_kernel": att["exact"]}
    gemms, tot_flop, tot_us = {}, 0.0, 0.0
    # (the shared expert is a slot of the grouped expert GEMMs, not a dense GEMM: MoEDeepSeekV3.forward)
    for name, (N, K) in {"wqkv_a": (2112, 7168), "wq_b": (3072, 1536), "wo": (7168, 2048), "dense_w1w3": (4608, 7168),
                         "dense_w2": (7168, 2304)}.items():
        x = (torch.randn(T, K, device="cuda", generator=gd)).to(torch.bfloat16)
        xq, xs = ops.act_quant_deepseek_v3(x)
        w = (torch.randn(N, K, device="cuda", generator=gd) * 0.5).to(torch.float8_e4m3fn)
        ws = torch.rand((N + 127) // 128, (K + 127) // 128, device="cuda", generator=gd) * 0.02 + 0.01
        us = time_us(lambda: ops.fp8_gemm_deepseek_v3(xq, xs, w, ws, out_dtype=torch.bfloat16))
        f = 2.0 * T * N * K
        gemms[name] = {"N": N, "K": K, "us": round(us, 1), "TFLOPs": round(f / us * 1e-6, 1), "frac": round(f / us * 1e-6 / FP8_MFMA_PEAK_TFLOPS, 4)}
        if not name.startswith("dense"):
            tot_flop += f
            tot_us += us
    out["fp8_gemm_tiled"] = {"kernel": "chitu::fp8_gemm_tiled_kernel (v_mfma_scale_f32_16x16x128_f8f6f4)", "bound": "mfma", "peak_TFLOPs": FP8_MFMA_PEAK_TFLOPS, "M": T,
                             "moe_layer_dense_gemms_TFLOPs": round(tot_flop / tot_us * 1e-6, 1),
                             "moe_layer_dense_gemms_frac": round(tot_flop / tot_us * 1e-6 / FP8_MFMA_PEAK_TFLOPS, 4), "shapes": gemms}
    return out


def ep8_rank_extra(steps, warmup, ctx, moe_rank=0):
    """ONE rank of an expert-parallel R1 deployment (the reference only stubs it, fused_moe.py:163-179,
    model_deepseek_v3.py:870-871, 1004) -- attention at 1/8 of the heads as in the TP=8 shard, routed experts
    [32*moe_rank, 32*moe_rank + 32) at FULL width, 1/8 of the shared expert's width; the same weight bytes per rank as the
    TP shard, and about the same bytes per step (the rank streams the local experts its tokens hit, whole).  The combine
    all-reduce is not paid on a lone GPU (as in the TP shard run)."""
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Args, DeepSeekV3Decoder, init_synthetic_

    margs = DeepSeekV3Args(shard_degree=SHARD, moe_world_size=SHARD, moe_rank=moe_rank)
    max_seq = ctx + steps + warmup + 256
    cache = PagedKVCacheManager(0, margs.n_layers, num_hot_req=16, block_size=64, max_seq_len=max_seq, device="cuda",
                                kv_shape_per_sample=(576,), dtype=torch.bfloat16)
    model = DeepSeekV3Decoder(margs, cache, HipAttnBackend(local_n_heads=margs.n_heads // SHARD, max_seq_len=max_seq),
                              max_position_embeddings=max(max_seq, 4097), device="cuda")
    init_synthetic_(model, seed=1000)
    cache.paged_kv_cache.normal_(0, 0.5)
    out = {"model": f"DeepSeek-R1 FP8, one expert-parallel rank (ep=8, rank {moe_rank}: 32 routed experts at full "
                    "width + 1/8 shared width, attention 1/8 heads), hipGraph, synthetic weights"}
    n_local = margs.n_routed_experts // SHARD
    lo = moe_rank * n_local
    d, t = margs.dim, SHARD
    attn = (margs.q_lora_rank + margs.kv_lora_rank + margs.qk_rope_head_dim) * d \
        + (margs.n_heads * 192 // t) * margs.q_lora_rank + (margs.n_heads * 256 // t) * margs.kv_lora_rank + d * (margs.n_heads * 128 // t)
    n_moe = margs.n_layers - margs.n_dense_layers
    for bs in EXTRA_BATCHES:
        dt = measure(model, cache, bs, ctx, steps, warmup, 1, True, f"e{bs}_")
        routing = capture_step_routing(model, cache, bs, ctx)
        # local routed experts a step streams (whole, 3 * 2048 * 7168 B each), measured on an eager step of this model
        local = sum(int(r[(r >= lo) & (r < lo + n_local)].unique().numel()) for r in routing) / max(1, len(routing))
        moe = margs.n_routed_experts * d * 2 + local * 3 * margs.moe_inter_dim * d + 3 * margs.moe_inter_dim * d // t
        alg = margs.n_layers * attn + margs.n_dense_layers * 3 * margs.inter_dim * d // t + n_moe * moe \
            + (margs.vocab_size // t) * d * 2 + margs.n_layers * bs * ctx * 576 * 2
        out[f"bs{bs}"] = {"ms_per_step": round(dt / steps * 1e3, 4), "node_tok_s": round(bs * steps / dt, 1),
                          "local_routed_experts_hit": round(local, 2), "step_algorithmic_GB": round(alg / 1e9, 3),
                          "roofline_frac": round(alg / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4)}
    del model, cache
    torch.cuda.empty_cache()
    return out


def mixtral_extra(steps, warmup, ctx):
    """BASELINE config 4 as an extra line: Mixtral-8x7B shapes, INT8 W8A8 experts (per-token / per-channel
    scales) through the grouped int8 MoE kernels, bf16 attention and router; the whole model on this GPU
    (the config's TP=4 needs 4 GPUs), paged KV (page 256), hipGraph, synthetic weights."""
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.mixtral import MixtralArgs, MixtralDecoder, init_synthetic_

    args = MixtralArgs()
    max_seq = ctx + steps + warmup + 512
    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=16, block_size=256, max_seq_len=max_seq, device="cuda",
                                n_local_kv_heads=args.n_kv_heads, head_dim=args.head_dim, dtype=torch.bfloat16)
    model = MixtralDecoder(args, cache, HipAttnBackend(local_n_heads=args.n_heads, max_seq_len=max_seq),
                           max_position_embeddings=max_seq, device="cuda")
    init_synthetic_(model, seed=4)
    cache.paged_k_cache.normal_(0, 0.5)
    cache.paged_v_cache.normal_(0, 0.5)
    out = {"model": "Mixtral-8x7B shapes, INT8 W8A8 experts + bf16 attention, TP=1, paged KV (page 256), hipGraph, synthetic weights"}
    d, E, k = args.dim, args.num_local_experts, args.num_experts_per_tok
    attn_w = ((args.n_heads + 2 * args.n_kv_heads) * args.head_dim * d + d * args.n_heads * args.head_dim) * 2  # bf16
    expert_w = 3 * args.ffn_dim * d  # int8: 1 B / weight
    head_w = args.vocab_size * d * 2
    for bs in EXTRA_BATCHES:
        dt = measure(model, cache, bs, ctx, steps, warmup, 1, True, f"x{bs}_")
        # experts a step streams: expected distinct ones under the synthetic router's near-uniform top-k (8 experts: a
        # batch of 16 hits all of them with probability 0.99 per expert)
        distinct = E * (1.0 - (1.0 - k / E) ** bs)
        alg = args.n_layers * (attn_w + E * d * 2 + distinct * expert_w) + head_w \
            + args.n_layers * bs * ctx * args.n_kv_heads * args.head_dim * 2 * 2
        out[f"bs{bs}"] = {"ms_per_step": round(dt / steps * 1e3, 4), "tok_s": round(bs * steps / dt, 1),
                          "expected_distinct_experts": round(distinct, 2), "step_algorithmic_GB": round(alg / 1e9, 3),
                          "roofline_frac": round(alg / (dt / steps) / 1e9 / HBM_PEAK_GBS, 4)}
    del model, cache
    torch.cuda.empty_cache()
    return out


def main():
    global NO_GRAPH_CHECK
    a = parse()
    NO_GRAPH_CHECK = a.no_graph_check
    rank, world, local, dinfo = setup_dist(a.gpus)
    if a.opt:  # A/B of launch variants on one box (tools): identical results, different kernels
        from chitu_amd import _lib

        for kv in a.opt.split(","):
            k, v = kv.split("=")
            _lib.check(_lib.lib().chitu_hip_debug_option(_lib.i32(_lib.DEBUG_OPTIONS[k]), _lib.i32(int(v))), "debug_option")
    use_graph = not a.no_graph
    if dinfo["shared_device"]:
        # all ranks on one GPU: keep the layers that fit (1.45 GB per layer and rank); flagged invalid below
        fit = int(torch.cuda.mem_get_info()[0] * 0.7 / world / 1.45e9)
        a.layers = max(4, min(a.layers, fit))
    t_build = time.perf_counter()
    margs, model, cache = build_model(a, rank)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build
    batches = [a.bs] + [b for b in ((1, 32) if not a.no_bs1 else ()) if b != a.bs]
    transport = enable_collectives(world, max(batches), model.vocab_local)

    # One rank, or N ranks on the xGMI collectives: the step is ONE hipGraph.  N ranks on the library: hipGraph
    # PIECES cut at every collective (DeepSeekV3Decoder.decode, "piecewise"); CHITU_TP_GRAPH=full asks for the
    # RCCL calls to be captured too -- that attempt is probed first and every rank falls back to eager launches
    # together if the capture is refused.
    from chitu_amd import graphs

    graph_mode = "off" if not use_graph else {"full": "on", "piecewise": "piecewise"}[graphs.graph_mode(True)]
    if use_graph and world > 1 and transport.startswith("nccl") and graph_mode == "on":
        ok = torch.ones(1, device="cuda")
        try:
            measure(model, cache, a.bs, a.ctx, 2, 1, world, True, "probe")
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] rank {rank}: graph capture failed ({type(exc).__name__}: {exc}); eager fallback", file=sys.stderr)
            torch.cuda.set_stream(torch.cuda.default_stream())  # the failed capture leaves its side stream current
            for _ in range(16):  # drain the sticky capture-invalidated errors before touching the device again
                try:
                    torch.cuda.synchronize()
                    ok = torch.zeros(1, device="cuda")
                    break
                except Exception:  # noqa: BLE001
                    pass
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if ok.item() == 0:
            use_graph, graph_mode = False, "off (full capture refused)"
            model.graphs, model.static_tokens, model.static_out, model.graph_pool = {}, {}, {}, None
    dt = measure(model, cache, a.bs, a.ctx, a.steps, a.warmup, world, use_graph, "m")
    ms_per_step = dt / a.steps * 1e3
    # the same loop once more, NOT part of any reported time, with rocm-smi sampled beside it
    with DeviceStateSampler(local if not dinfo["shared_device"] else 0) as dev_state:
        measure(model, cache, a.bs, a.ctx, max(a.steps, 96), 0, world, use_graph, "d")
    node_tok_s = a.bs * a.steps / dt
    value = node_tok_s * world / SHARD

    extra = {}
    for b in batches[1:]:  # BASELINE config 5: bs in {1, 16, 32}
        dtb = measure(model, cache, b, a.ctx, a.steps, a.warmup, world, use_graph, f"s{b}_")
        extra[f"bs{b}"] = {
            "ms_per_step": round(dtb / a.steps * 1e3, 4), "node_tok_s": round(b * a.steps / dtb, 2),
            "value": round(b * a.steps / dtb * world / SHARD, 3),
        }
    coll = None
    if world > 1:
        n_coll = 2 * margs.n_layers + 2
        coll = {"collectives_in_step": n_coll, "transport": transport, "library_backend": dinfo["backend"],
                "ranks_seen_by_library": dinfo["ranks_seen"],
                "fused": "every per-layer all-reduce runs inside the launch that also does the top-k sum, the residual "
                         "add, the RMSNorm and the fp8 quant" if transport.startswith("xgmi") else "no"}
        timing = time_collectives(a.bs, margs.dim, model.vocab_local)
        coll.update(timing)
        if timing.get("allreduce_add_norm_quant_us") and timing.get("logits_all_gather_us"):
            coll["collective_ms_per_step_est"] = round(
                ((n_coll - 1) * timing["allreduce_add_norm_quant_us"] + timing["logits_all_gather_us"]) * 1e-3, 3)
        from chitu_amd import tensor_parallel as tp

        if tp.xgmi_comm() is not None:
            coll["xgmi_error_word"] = tp.xgmi_comm().status()
        # the transport decision of enable_xgmi() on rank 0, the stage it reached, why it fell back if it did, and its
        # pre-flight: per-transport us at bs 1 / 16 / 32 measured eagerly BEFORE the first capture
        coll["xgmi_enable"] = dict(tp.xgmi_report) if os.environ.get("CHITU_ALLREDUCE", "xgmi") != "rccl" else {
            "enabled": False, "stage": "not attempted", "reason": "CHITU_ALLREDUCE=rccl"}
        coll["graph_form"] = ("one hipGraph per step with the collectives inside" if transport.startswith("xgmi") and not tp.xgmi_split_phase()
                              else "eager launches (split-phase collectives carry a host barrier)" if tp.xgmi_split_phase()
                              else "piecewise graph replay around the library's collectives (graphs.py)")
    else:
        coll = {"collectives_in_step": 0, "transport": transport,
                "note": "one rank of eight: the 124 collectives of the TP=8 step are not paid here"}

    roof = None
    if not a.no_roofline:
        # the routing capture runs decode steps, i.e. collectives at N > 1: every rank takes part;
        # the kernel timing itself is rank 0's
        routing = capture_step_routing(model, cache, a.bs, a.ctx)
        if rank == 0:
            roof = roofline_dominant_kernel(model, routing, margs, a.bs)
        if world > 1:
            dist.barrier()
    calib = None
    if rank == 0 and not a.no_calibration:
        w2_us = None
        for k in (roofline_array(roof) or []):
            if "gemm2" in str(k.get("kernel", "")):
                w2_us = k.get("avg_launch_us")
        calib = box_calibration(local if not dinfo["shared_device"] else 0, w2_us)
        try:
            calib["tail_launch_us"] = tail_launch_probe(model, cache, a.bs, a.ctx)
            # fast-box reference values of the same probe (profiles/r06_bench_default_*.json); > 1.10 x = the slow-box signature
            ref = CAL_REF.get("tail_launch_us", {})
            calib["tail_launch_vs_reference_box"] = {k: round(v["us_per_launch"] / ref[k], 3) for k, v in calib["tail_launch_us"].items()
                                                     if k in ref and "us_per_launch" in v}
        except Exception as exc:  # noqa: BLE001 -- diagnostics only
            calib["tail_launch_us"] = f"unavailable: {type(exc).__name__}: {exc}"[:200]
        calib["static_device_facts"] = static_device_facts(local if not dinfo["shared_device"] else 0)
    if world > 1:
        dist.barrier()
    distinct = (roof["distinct_experts"] - 1) if roof else min(256, a.bs * 8)  # routed only; shared counted in the formula
    step_bytes = algorithmic_bytes_per_step(margs, a.bs, a.ctx, distinct)
    if world == 1 and not a.no_llama and a.layers == 61:
        del model, cache
        model = cache = None
        torch.cuda.empty_cache()
        extra["llama3_8b"] = llama3_8b_extra(a.steps, a.warmup, a.ctx)
        extra["v2_lite"] = v2_lite_extra(a.steps, a.warmup, a.ctx)
        extra["mixtral_8x7b_int8"] = mixtral_extra(a.steps, a.warmup, a.ctx)
        extra["ep8_rank"] = ep8_rank_extra(a.steps, a.warmup, a.ctx)
        extra["prefill"] = prefill_extra()
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        model = cache = None
        torch.cuda.empty_cache()
        cpu = cpu_baseline(margs, a.bs, a.ctx)

    graph_rep = graph_report(world)  # (a collective at N > 1: every rank calls it)
    if world > 1:
        dist.barrier()
    if rank == 0:
        res = {
            "metric": f"output tok/s (bs={a.bs}) DeepSeek-R1 FP8 TP=8; HBM GB/s vs peak",
            "value": round(value, 3), "unit": "tok/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp8_e4m3 weights+activations (W8A8, fp32 accumulate), bf16 attention",
            "data": "synthetic (random-init weights, random KV cache, random prompts)",
            "config": {
                "workload": f"DeepSeek-R1-671B FP8 decode, one TP=8 rank shard per GPU "
                            f"({margs.n_layers} layers, 16 heads, 257 experts x 1/8 width), {world} of 8 shards live, "
                            f"bs={a.bs}, ctx={a.ctx}, greedy, hipGraph={graph_mode}",
                "batch": a.bs, "context": a.ctx, "parallelism": f"tp8-shard x{world}", "layers": margs.n_layers,
            },
            "value_is": "the N/8 share of node_tok_s (N of the node's 8 rank shards live); collectives between the live ranks only "
                        "(none at N=1): only N=8 is the TP=8 metric itself",
            "node_tok_s": round(node_tok_s, 2),
            "box_calibration": calib,
            "value_normalised": round(value * calib["step_time_factor"], 3) if calib else None,
            "value_normalised_is": "value x box_calibration.step_time_factor: what the reference box of the calibration constants "
                                   "would be expected to measure (two-term model; the measured `value` stands)",
            "collectives": coll,
            "step_algorithmic_GB": round(step_bytes / 1e9, 3),
            "step_hbm_GBs": round(step_bytes / (ms_per_step * 1e-3) / 1e9, 1),
            "step_roofline_frac": round(step_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": roof, "roofline_kernels": roofline_array(roof), "cpu_baseline": cpu, "build_s": round(build_s, 1),
            "graph_verified": graph_rep,
            "device_state_under_load": dev_state.summary(),
        }
        res.update(extra)
        if a.opt:
            res["launch_variant_overrides"] = a.opt
        if use_graph and (res["graph_verified"]["all_equal_eager"] is not True
                          or not res["graph_verified"]["captures_clean_on_every_rank"]):
            res["invalid"] = ("a timed hipGraph did not reproduce the eager launches of the same step, was not checked, or a "
                              "capture had to be rejected and repeated on some rank (graph_verified): the line is void")
            res["value"] = None
        elif dinfo["ranks_seen"] != a.gpus:
            # the library's own all-reduce of ones must have seen every rank the line claims
            res["invalid"] = f"--gpus {a.gpus} but the process group's all-reduce saw {dinfo['ranks_seen']} rank(s): the line is void"
            res["value"] = None
        elif coll and coll.get("xgmi_error_word"):
            res["invalid"] = "an xGMI collective timed out (error word != 0): the step's results and timings are void"
        elif a.layers != 61 or a.router_std is not None or dinfo["shared_device"]:
            res["invalid"] = ("debug run: " + ", ".join(
                w for w, c in (("reduced layer count", a.layers != 61), ("non-default synthetic router", a.router_std is not None),
                               ("all ranks share one GPU (functional check of the N > 1 path)", dinfo["shared_device"])) if c))
        print(json.dumps(res))
    if world > 1:
        from chitu_amd import tensor_parallel as tp

        torch.cuda.synchronize()
        dist.barrier()
        tp.disable_xgmi()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
