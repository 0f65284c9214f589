"""Root-cause tool for the round-3 hipGraph replay miscompare (GPUTEST_r03: the Llama reference run's 64 replayed decode
steps differ from the same steps launched eagerly, only inside a whole-suite process).

Experiments, one process, all on the tiny reference-Llama fixture (tests/golden/ref_llama.npz):
  E1  baseline: graph rows vs eager rows on a fresh model.
  E2  every torch.empty / new_empty / empty_like / workspace buffer poisoned with 0xFF bytes (NaN in bf16 / fp32 / fp8,
      -1 in int32), eager launches only: any kernel that READS memory nobody wrote shows up as a changed row.
  E3  device memory dirtied before the capture: many small and a few large allocations filled with a pattern, freed,
      cache emptied -- so the private pool of the capture is served recycled, non-zero memory -- then graph vs eager.
  E4  when a graph run differs: per-op outputs of one replay against one eager step on the same state; first op that
      differs, and how.
"""

import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from chitu_amd import ops, workspace  # noqa: E402
from chitu_amd.attn_backend import HipAttnBackend  # noqa: E402
from chitu_amd.cache_manager import PagedKVCacheManager  # noqa: E402
from chitu_amd.llama import LlamaArgs, LlamaDecoder  # noqa: E402
from tests.util import poisoned_allocations, ref_llama_fixture  # noqa: E402

G, CFG, P = ref_llama_fixture()
PROMPT, TOKS = G["prompt"].tolist(), G["tokens"].tolist()


def make():
    ffn = P["layers.0.ffn.w2"].shape[1]
    args = LlamaArgs(dim=CFG["dim"], n_layers=CFG["n_layers"], n_heads=CFG["n_heads"], n_kv_heads=CFG["n_kv_heads"],
                     vocab_size=CFG["vocab_size"], ffn_dim=ffn, norm_eps=CFG["norm_eps"], rope_theta=CFG["rope_theta"])
    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=1, block_size=256, max_seq_len=512, device="cuda",
                                n_local_kv_heads=args.n_kv_heads, head_dim=args.head_dim, dtype=torch.bfloat16)
    model = LlamaDecoder(args, cache, HipAttnBackend(local_n_heads=args.n_heads, max_seq_len=512),
                         max_position_embeddings=512, device="cuda")
    params = dict(model.named_parameters())
    for k, t in P.items():
        params[k].data.copy_(t)
    return model, cache


def run(model, cache, req, use_graph, steps=64):
    rows = [model.prefill([PROMPT], [req]).float().cpu()]
    tok = torch.tensor([TOKS[0]], dtype=torch.int64, device="cuda")
    for step in range(steps):
        cache.prepare_cache_decode([req])
        cache.prepare_block_table_for_decode([req])
        rows.append(model.decode(tok, use_graph=use_graph).float().cpu())
        cache.finalize_cache_single_decode([req])
        tok = torch.tensor([TOKS[step + 1]], dtype=torch.int64, device="cuda")
    cache.finalize_cache_all_decode(req)
    return torch.cat(rows)


def ref_err(rows):
    ref = torch.from_numpy(G["logits"])[: rows.shape[0]]
    return ((rows - ref).abs().amax(-1) / ref.abs().amax(-1)).max().item()


# ---------------------------------------------------------------- E2: poisoned allocations (tests/util.py)


# ---------------------------------------------------------------- E3: dirty device memory
def dirty_device(pattern: str, small_n=4000, large_gb=8):
    """Fill recycled device memory with non-zero bytes and hand it back to the runtime (not to torch's cache)."""
    keep = []
    gen = torch.Generator(device="cuda").manual_seed(7)
    sizes = [512, 4096, 65536, 1 << 19, 1 << 20, 3 << 20, 20 << 20]
    for i in range(small_n):
        t = torch.empty(sizes[i % len(sizes)], dtype=torch.uint8, device="cuda")
        keep.append(t)
    for _ in range(large_gb):
        keep.append(torch.empty(1 << 30, dtype=torch.uint8, device="cuda"))
    for t in keep:
        if pattern == "ff":
            t.fill_(0xFF)
        elif pattern == "rand":
            t.random_(0, 256, generator=gen)
        elif pattern == "huge":  # bf16 0x7F00 = 1.7e38, finite
            t.view(torch.int16).fill_(0x7F00)
        elif pattern == "one":
            t.view(torch.int16).fill_(0x3F80)  # bf16 1.0 / int32 0x3F803F80
    torch.cuda.synchronize()
    del keep
    torch.cuda.empty_cache()


# ---------------------------------------------------------------- E4: per-op comparison of one replay and one eager step
REC = None


def _wrap(mod, name):
    real = getattr(mod, name)

    def f(*a, **k):
        out = real(*a, **k)
        if REC is not None:
            outs = out if isinstance(out, (tuple, list)) else (out,)
            REC.append((name, [o.clone() for o in outs if isinstance(o, torch.Tensor)]))
        return out

    setattr(mod, name, f)
    return real


def per_op_diff():
    global REC
    names = ["embed_rope_gather", "bf16_linear", "bf16_linear_add_norm_qkv_post", "bf16_linear_silu_add_norm", "rms_norm",
             "bf16_linear_add_norm", "bf16_linear_silu", "gqa_qkv_post"]
    reals = {n: _wrap(ops, n) for n in names if hasattr(ops, n)}
    real_attn = HipAttnBackend.attn_with_kvcache

    def attn(self, *a, **k):
        out = real_attn(self, *a, **k)
        if REC is not None:
            REC.append(("attn_with_kvcache", [out.clone()]))
        return out

    HipAttnBackend.attn_with_kvcache = attn
    try:
        model, cache = make()
        req = "p"
        model.prefill([PROMPT], [req])
        tok = torch.tensor([TOKS[0]], dtype=torch.int64, device="cuda")
        cache.prepare_cache_decode([req])
        cache.prepare_block_table_for_decode([req])
        REC = []
        out_g = model.decode(tok, use_graph=True).clone()  # pre-run + capture record into REC; then one replay
        torch.cuda.synchronize()
        n_ops = len(REC) // 2  # the eager pre-run's clones, then the captured clones (rewritten by the replay)
        pre, cap = REC[:n_ops], REC[n_ops:]
        REC = []
        out_e = model.decode(tok, use_graph=False).clone()
        torch.cuda.synchronize()
        eag, REC = REC, None
        print(f"  per-op: {n_ops} ops; logits replay==eager {torch.equal(out_g, out_e)}")
        first = None
        for i, ((n1, a), (n2, b), (n3, c)) in enumerate(zip(pre, cap, eag)):
            same_gc = all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(b, c))
            same_pe = all(torch.equal(x.view(torch.uint8), y.view(torch.uint8)) for x, y in zip(a, c))
            if not (same_gc and same_pe) and first is None:
                first = i
            if not (same_gc and same_pe):
                d = [(x.float() - y.float()).abs().max().item() for x, y in zip(b, c)]
                nan = [bool(torch.isnan(x.float()).any()) for x in b]
                print(f"    op {i} {n2}: replay==eager {same_gc}, pre-run==eager {same_pe}, max diff {d}, replay has NaN {nan}")
        print("  first differing op:", first)
        return first is None
    finally:
        for n, r in reals.items():
            setattr(ops, n, r)
        HipAttnBackend.attn_with_kvcache = real_attn


def graph_vs_eager(tag):
    model, cache = make()
    rg = run(model, cache, tag + "g", True)
    re_ = run(model, cache, tag + "e", False)
    same = int((rg == re_).all(-1).sum())
    print(f"  [{tag}] graph rows == eager rows: {same} / {rg.shape[0]}; vs reference: graph {ref_err(rg):.4g} eager {ref_err(re_):.4g}; "
          f"reserved {torch.cuda.memory_reserved() / 2**20:.0f} MB")
    return same == rg.shape[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--patterns", default="ff,rand,huge,one")
    ap.add_argument("--repeats", type=int, default=2)
    a = ap.parse_args()
    print("E1 baseline")
    ok = graph_vs_eager("base")
    model, cache = make()
    base = run(model, cache, "b", False)
    print("E2 poisoned allocations, eager")
    with poisoned_allocations():
        model2, cache2 = make()
        pois = run(model2, cache2, "p", False)
    same = int((base == pois).all(-1).sum())
    print(f"  eager rows identical with poisoned allocations: {same} / {base.shape[0]}; NaN rows {int(torch.isnan(pois).any(-1).sum())}")
    print("E2b poisoned allocations inside the capture too")
    with poisoned_allocations():
        model3, cache3 = make()
        pg = run(model3, cache3, "pg", True)
    print(f"  graph rows (poison fills captured) identical to clean eager: {int((base == pg).all(-1).sum())} / {base.shape[0]}")
    del model, cache, model2, cache2, model3, cache3
    bad = []
    for pat in a.patterns.split(","):
        for r in range(a.repeats):
            print(f"E3 dirty device memory, pattern {pat}, repeat {r}")
            dirty_device(pat)
            if not graph_vs_eager(f"{pat}{r}"):
                bad.append((pat, r))
                print("E4 per-op comparison in this state")
                dirty_device(pat)
                per_op_diff()
    print("E4 per-op comparison, clean state")
    per_op_diff()
    print("SUMMARY: baseline ok", ok, "| dirty-memory failures:", bad)


if __name__ == "__main__":
    main()
