"""Second root-cause experiment for the hipGraph replay miscompare: does a NEW graph run on the kernel arguments of a
DESTROYED one?

graph_repro.py could not see that: every instance there had the same weights and inputs, so a replay on stale
arguments (pointers into the freed -- but intact -- tensors of the previous instance) returns the right numbers.  Here
the previous instances ("decoys") carry OTHER weights and token streams; the instance under test carries the fixture's.

Per trial: build D decoy models (random weights), capture + replay each one's decode graph a few steps, destroy them
(graphs, private pools, weights), then build the fixture model and compare its 64 replayed steps with the same steps
launched eagerly and with the reference rows.
"""

import argparse
import gc
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from tools.graph_repro import G, PROMPT, TOKS, make, ref_err, run  # noqa: E402


def decoy(seed, steps):
    model, cache = make()
    gen = torch.Generator(device="cuda").manual_seed(seed)
    for p in model.parameters():
        p.data.copy_((torch.randn(p.shape, device="cuda", generator=gen) * (p.shape[-1] ** -0.5)).to(p.dtype))
    req = f"d{seed}"
    prompt = [(7 * seed + i) % 1000 for i in range(5 + seed % 7)]
    model.prefill([prompt], [req])
    tok = torch.tensor([seed % 1000], dtype=torch.int64, device="cuda")
    out = None
    for s in range(steps):
        cache.prepare_cache_decode([req])
        cache.prepare_block_table_for_decode([req])
        out = model.decode(tok, use_graph=True)
        cache.finalize_cache_single_decode([req])
        tok = out.argmax(-1)
    torch.cuda.synchronize()
    return model, cache


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--trials", type=int, default=6)
    ap.add_argument("--decoys", type=int, default=4)
    ap.add_argument("--keep", type=int, default=0, help="decoys kept alive across the test instance")
    ap.add_argument("--empty-cache", type=int, default=1)
    a = ap.parse_args()
    knobs = {k: os.environ[k] for k in ("DEBUG_CLR_KERNARG_HDP_FLUSH_WA", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "HIP_FORCE_DEV_KERNARG",
                                        "GPU_MAX_HW_QUEUES") if k in os.environ}
    print("knobs:", knobs, "args:", vars(a))
    bad = 0
    for t in range(a.trials):
        alive = [decoy(100 * t + d, 3 + d) for d in range(a.decoys)]
        kept = alive[: a.keep]
        del alive
        gc.collect()
        if a.empty_cache:
            torch.cuda.empty_cache()
        model, cache = make()
        rg = run(model, cache, f"t{t}g", True)
        re_ = run(model, cache, f"t{t}e", False)
        same = int((rg == re_).all(-1).sum())
        print(f"trial {t}: graph rows == eager rows {same} / {rg.shape[0]}; vs reference graph {ref_err(rg):.4g} eager {ref_err(re_):.4g}")
        if same != rg.shape[0]:
            bad += 1
            # what does the bad graph do after the caches were churned / a device-wide sync?
            torch.cuda.synchronize()
            junk = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
            for _ in range(4):
                junk.fill_(1)
            torch.cuda.synchronize()
            del junk
            rg2 = run(model, cache, f"t{t}g2", True)  # same graph object, replayed on a new request
            print(f"   after 4 GB of fills: graph rows == eager rows {int((rg2 == re_).all(-1).sum())} / {rg.shape[0]}")
            model.graphs.clear()
            rg3 = run(model, cache, f"t{t}g3", True)  # re-captured on the same model
            print(f"   re-captured on the same model: graph rows == eager rows {int((rg3 == re_).all(-1).sum())} / {rg.shape[0]}")
        del model, cache, kept
        gc.collect()
    print("BAD TRIALS:", bad, "of", a.trials)


if __name__ == "__main__":
    main()
