"""Hunt for the launch whose result depends on timing: the bs-32 decode step of a 12-layer R1 rank shard, eager, repeated on the
same state WHILE a second process keeps the GPU busy (the N = 2 shared-GPU bench failed its graph == eager check at bs 32 on the
round-5 tree and on this one, on the library path too: profiles/r06_n2_graph_check.txt).  Prints, per repetition, the first layer /
sub-module whose output differs from repetition 0 and which rows.   python tools/r06_race_hunt.py [bs=32] [layers=12] [reps=12] [noise=1]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

NOISE = r'''
import torch, time
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
small = torch.zeros(64, device="cuda")
t0 = time.time()
while time.time() - t0 < float(%f):
    for _ in range(20):
        b = a @ a
        big.add_(1)
        for _ in range(30):
            small.add_(1.0)
    torch.cuda.synchronize()
'''


def flat(o):
    if isinstance(o, torch.Tensor):
        return [o]
    if isinstance(o, (list, tuple)):
        return [t for x in o for t in flat(x)]
    if hasattr(o, "q") and hasattr(o, "s"):
        return [o.q.view(torch.uint8), o.s]
    if hasattr(o, "resolve"):
        return flat(getattr(o, "part", None))
    return []


@torch.inference_mode()
def main():
    bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    layers = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    noise = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    poison = os.environ.get("HUNT_POISON", "0") == "1"
    torch.cuda.set_device(0)
    if world > 1:  # as bench.py --gpus N on a box with one GPU: every rank on cuda:0, library collectives through gloo
        import torch.distributed as dist
        from chitu_amd import tensor_parallel as tp

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        tp.init_tp(world, 1)
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Args, DeepSeekV3Decoder, init_synthetic_

    margs = DeepSeekV3Args(shard_degree=8, n_layers=layers)
    cache = PagedKVCacheManager(0, layers, num_hot_req=bs, block_size=64, max_seq_len=1024 + 64, device="cuda",
                                kv_shape_per_sample=(576,), dtype=torch.bfloat16)
    model = DeepSeekV3Decoder(margs, cache, HipAttnBackend(local_n_heads=16, max_seq_len=1024 + 64), max_position_embeddings=4097, device="cuda")
    init_synthetic_(model, seed=1000 + rank)
    gen = torch.Generator(device="cuda").manual_seed(77)
    cache.paged_kv_cache.view(-1).copy_(torch.randn(cache.paged_kv_cache.numel(), device="cuda", dtype=torch.bfloat16, generator=gen) * 0.5)
    reqs = [f"r{i}" for i in range(bs)]
    for r in reqs:
        cache.register_sequence(r, 1024)
    tokens = torch.randint(100, 1000, (bs,), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    cache.prepare_cache_decode(reqs)
    cache.prepare_block_table_for_decode(reqs)
    rec = []
    names = {}
    from chitu_amd import deepseek_v3 as dsv3, ops as _ops, tensor_parallel as _tp

    def wrap(obj, name, label):
        fn = getattr(obj, name)

        def traced(*a, **k):
            if "absorb_bmm" in label:
                rec.append((label + ":pre", [t.detach().clone() for t in flat([x for x in a if not isinstance(x, torch.nn.Module)])]))
            o = fn(*a, **k)
            ins = [t.detach().clone() for t in flat([x for x in a if not isinstance(x, torch.nn.Module)]) if t.numel() < (1 << 22)]  # (not the KV cache)
            rec.append((label + ":in", ins))
            rec.append((label, [t.detach().clone() for t in flat(o)]))
            return o

        setattr(obj, name, traced)

    wrap(_ops, "embed_rope_gather", "ops.embed_rope_gather")
    wrap(_tp, "all_reduce", "tp.all_reduce")
    wrap(_tp, "add_norm", "tp.add_norm")
    wrap(_tp, "all_gather_last_dim", "tp.all_gather_last_dim")
    wrap(dsv3.AttentionDeepSeekV3, "decode_forward_paged", "attn.decode_forward_paged")
    wrap(_ops, "rms_norm", "ops.rms_norm")
    from chitu_amd.attn_backend import HipAttnBackend as _HB
    for fn_name in ("mla_q_proj", "absorb_bmm_rope_fp8", "mla_merge_absorb_uv_quant_fp8", "fp8_gemm_deepseek_v3", "mla_qkv_post",
                    "absorb_uv_quant_fp8", "fp8_linear_add_norm"):
        if hasattr(_ops, fn_name):
            wrap(_ops, fn_name, "ops." + fn_name)
    wrap(_HB, "mla_decode", "HipAttnBackend.mla_decode")
    for n, m in model.named_modules():
        if n and n.count(".") <= 2:
            names[m] = n
            m.register_forward_hook(lambda mod, i, o: rec.append((names[mod], [t.detach().clone() for t in flat(o)])))
    proc = None
    if noise:
        proc = subprocess.Popen([sys.executable, "-c", NOISE % 600.0])
        time.sleep(8)
    ref_out, ref_rec = None, None
    bad_runs = 0
    for rep in range(reps):
        rec.clear()
        if poison and rep > 0:
            # every cached free block of the allocator gets NaN patterns: a launch that reads memory it (or a producer) did not
            # write this step now reads NaN instead of last step's -- identical -- values
            junk = [torch.full((n,), float("nan"), dtype=torch.bfloat16, device="cuda") for n in
                    [1 << k for k in range(8, 27)] + [3 * (1 << k) for k in range(8, 25)] + [32 * 7168, 32 * 9 * 7168, 32 * 2112, 32 * 3072, 16 * 7168]]
            torch.cuda.synchronize()
            del junk
        out = model.decode(tokens, use_graph=False).clone()
        torch.cuda.synchronize()
        if rep == 0:
            ref_out, ref_rec = out, list(rec)
            print(f"rep 0: {len(rec)} hooked outputs", flush=True)
            continue
        if torch.equal(out, ref_out):
            continue
        bad_runs += 1
        rows = (out != ref_out).any(-1).nonzero().flatten().tolist()
        first = None
        for (n0, t0), (n1, t1) in zip(ref_rec, rec):
            assert n0 == n1
            for k, (a, b) in enumerate(zip(t0, t1)):
                if a.shape == b.shape and not torch.equal(a, b):
                    d = (a != b)
                    r = d.reshape(d.shape[0], -1).any(-1).nonzero().flatten().tolist() if d.dim() > 1 else []
                    first = (n0, k, tuple(a.shape), r[:16], int(d.sum()))
                    break
            if first:
                break
        if first and first[0].startswith("ops.absorb_bmm_rope_fp8:in") and rank == 0:
            pairs = [(i, n0) for i, ((n0, t0), (n1, t1)) in enumerate(zip(ref_rec, rec)) if n0 == "ops.absorb_bmm_rope_fp8:in"
                     and any(a.shape == b.shape and not torch.equal(a, b) for a, b in zip(t0, t1))]
            i = pairs[0][0]
            k3 = next(k for k, t in enumerate(ref_rec[i][1]) if tuple(t.shape[-2:]) == (16, 64))
            post0, post1 = ref_rec[i][1][k3], rec[i][1][k3]
            # the ":pre" record of the same call sits two entries earlier (pre, in, out)
            j = max(k for k in range(i) if ref_rec[k][0] == "ops.absorb_bmm_rope_fp8:pre")
            k4 = next(k for k, t in enumerate(ref_rec[j][1]) if tuple(t.shape[-2:]) == (16, 64))
            pre0, pre1 = ref_rec[j][1][k4], rec[j][1][k4]
            d = (post0 != post1).nonzero()
            print(f"  q_pe before the launch equal in both repetitions: {torch.equal(pre0, pre1)}; after it differs at (row, head, col): {d[:16].tolist()}")
            fl = [t for t in ref_rec[j][1] if t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 32]
            fl1 = [t for t in rec[j][1] if t.dtype == torch.float32 and t.dim() == 2 and t.shape[1] == 32]
            print(f"  cos / sin equal in both repetitions: {[bool(torch.equal(a_, b_)) for a_, b_ in zip(fl, fl1)]} (shapes {[tuple(t.shape) for t in fl]})")
            if len(fl) >= 2:
                cos_, sin_ = fl[-2], fl[-1]
                for r_, h_, c_ in d[:6].tolist():
                    i_ = c_ // 2
                    x0, x1 = pre0[r_, h_, 2 * i_].float(), pre0[r_, h_, 2 * i_ + 1].float()
                    e0 = (x0 * cos_[r_, i_] - x1 * sin_[r_, i_]).to(torch.bfloat16).item()
                    e1 = (x1 * cos_[r_, i_] + x0 * sin_[r_, i_]).to(torch.bfloat16).item()
                    print(f"    pair {i_} of ({r_},{h_}): expected ({e0:.5f}, {e1:.5f}); rep0 ({post0[r_, h_, 2*i_].item():.5f}, {post0[r_, h_, 2*i_+1].item():.5f}); this ({post1[r_, h_, 2*i_].item():.5f}, {post1[r_, h_, 2*i_+1].item():.5f})")
            for r_, h_, c_ in d[:6].tolist():
                print(f"    ({r_},{h_},{c_}): before {pre0[r_, h_, c_].item():.5f} / {pre1[r_, h_, c_].item():.5f}   after: rep0 {post0[r_, h_, c_].item():.5f}  this {post1[r_, h_, c_].item():.5f}")
        print(f"rank {rank} rep {rep}: nan in logits: {bool(torch.isnan(out).any())}; logits differ in rows {rows}; first differing module output: {first}", flush=True)
    print(f"rank {rank}: {bad_runs} of {reps - 1} repetitions differ from repetition 0 (noise process: {noise}, poison: {poison}, world {world})", flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if proc is not None:
        proc.kill()
        proc.wait()


main()
