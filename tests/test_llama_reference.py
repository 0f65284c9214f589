"""BASELINE config 1 -- "Llama-2-7B bf16 TP=1 decode on the reference CPU path, bs=1, 64 new tokens (plumbing, no
GPU)" -- as a parity case at test size: the reference's own TransformerLlama run on CPU (tests/golden/gen_ref_llama.py
-> ref_llama.npz) against the oracle's Llama composition here, and against the HIP LlamaDecoder on the GPU."""

import numpy as np
import pytest
import torch

from oracle import llama as ollama
from tests.util import ref_llama_fixture


def _check(logits, g, n_prompt, what):
    """logits [n_prompt + 64, vocab] of a teacher-forced run vs the fixture's 65 rows (prefill's last + 64 steps)."""
    ref = torch.from_numpy(g["logits"])
    mine = logits[n_prompt - 1:]
    assert mine.shape == ref.shape
    per_row = (mine - ref).abs().amax(-1) / ref.abs().amax(-1)
    err = per_row.max().item()
    assert err < 2e-2, (what, err, "rows over the bar:", (per_row >= 2e-2).nonzero().flatten().tolist(),
                        "all-zero rows:", (mine.abs().amax(-1) == 0).nonzero().flatten().tolist(),
                        "zero fraction / peak of the worst row:", (mine[int(per_row.argmax())] == 0).float().mean().item(),
                        mine[int(per_row.argmax())].abs().max().item(), "identical consecutive rows:",
                        int((mine[1:] == mine[:-1]).all(-1).sum()))
    top2 = ref.topk(2, -1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.05 * ref.abs().amax(-1)  # rows whose greedy choice is not a near-tie
    agree = mine.argmax(-1) == torch.from_numpy(g["tokens"])
    assert clear.sum() >= 30 and bool(agree[clear].all()), (what, int(clear.sum()), int(agree.sum()))
    return err, int(agree.sum())


def _forced_launch_variants(_lib):
    """Launch-variant overrides (chitu_hip_debug_option) still in force: name -> value; {} when every one is at -1."""
    import ctypes

    try:
        arr = (ctypes.c_int * len(_lib.DEBUG_OPTIONS)).in_dll(_lib.lib(), "_ZN5chitu15g_debug_optionsE")
        return {k: arr[i] for k, i in _lib.DEBUG_OPTIONS.items() if arr[i] != -1}
    except Exception as e:  # the symbol is an implementation detail: a diagnostic must not turn into the failure
        return repr(e)


def test_oracle_llama_reproduces_the_reference_cpu_run():
    g, cfg, p = ref_llama_fixture()
    prompt, toks = g["prompt"].tolist(), g["tokens"].tolist()
    fed = prompt + toks[:-1]  # teacher forcing: the reference's own greedy tokens
    hd = cfg["dim"] // cfg["n_heads"]
    logits = ollama.decode_sequence(p, fed, cfg["n_layers"], cfg["n_heads"], cfg["n_kv_heads"], hd, cfg["norm_eps"],
                                    cfg["rope_theta"], page=64)
    err, agree = _check(logits, g, len(prompt), "oracle")
    assert agree >= 63  # 65 greedy tokens; a bf16 near-tie may flip
    print("oracle vs reference Llama: rel err", err, "greedy agreement", agree, "/ 65")


@pytest.mark.gpu
def test_hip_llama_reproduces_the_reference_cpu_run():
    """Prefill of the prompt, then 64 hipGraph decode steps fed the reference's tokens: logits within 2e-2 of the
    reference's CPU run, identical greedy tokens wherever the reference's choice is not a near-tie."""
    from chitu_amd import _lib, graphs
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.llama import LlamaArgs, LlamaDecoder

    g, cfg, p = ref_llama_fixture()
    prompt, toks = g["prompt"].tolist(), g["tokens"].tolist()
    ffn = p["layers.0.ffn.w2"].shape[1]
    args = LlamaArgs(dim=cfg["dim"], n_layers=cfg["n_layers"], n_heads=cfg["n_heads"], n_kv_heads=cfg["n_kv_heads"],
                     vocab_size=cfg["vocab_size"], ffn_dim=ffn, norm_eps=cfg["norm_eps"], rope_theta=cfg["rope_theta"])

    def attempt(tag):
        """A fresh cache + model; the 65 logits rows of the graph run and of the same steps as eager launches."""
        cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=1, block_size=256, max_seq_len=512, device="cuda",
                                    n_local_kv_heads=args.n_kv_heads, head_dim=args.head_dim, dtype=torch.bfloat16)
        model = LlamaDecoder(args, cache, HipAttnBackend(local_n_heads=args.n_heads, max_seq_len=512),
                             max_position_embeddings=512, device="cuda")
        params = dict(model.named_parameters())
        assert set(params) == set(p)
        for k, t in p.items():
            assert params[k].shape == t.shape, k
            params[k].data.copy_(t)

        def run(req, use_graph):
            rows = [model.prefill([prompt], [req]).float().cpu()]
            tok = torch.tensor([toks[0]], dtype=torch.int64, device="cuda")
            for step in range(64):
                cache.prepare_cache_decode([req])
                cache.prepare_block_table_for_decode([req])
                if step in (0, 1, 63):  # the device-side step state the kernels will read == the host's bookkeeping
                    n_pages = len(cache.block_table[req])
                    assert cache.get_gpu_block_table()[0, :n_pages].tolist() == list(cache.block_table[req]), (req, step, "block table")
                    assert cache.get_gpu_seq_lens_excl_this_decode().tolist() == [cache.seq_lens[req]], (req, step, "lens excl")
                    assert cache.get_gpu_seq_lens_incl_this_decode().tolist() == [cache.seq_lens[req] + 1], (req, step, "lens incl")
                rows.append(model.decode(tok, use_graph=use_graph).float().cpu())
                cache.finalize_cache_single_decode([req])
                tok = torch.tensor([toks[step + 1]], dtype=torch.int64, device="cuda")
            cache.finalize_cache_all_decode(req)
            return rows

        return run(tag + "g", True), run(tag + "e", False)  # eager too: tells a replay problem from a kernel problem

    def verdict(rows):
        logits = torch.cat([torch.zeros(len(prompt) - 1, rows[0].shape[-1])] + rows)
        try:
            return _check(logits, g, len(prompt), "hip"), None
        except AssertionError as e:
            return None, e

    rows, rows_eager = attempt("a")
    same = sum(int(torch.equal(a, b)) for a, b in zip(rows, rows_eager))
    print("graph replay rows identical to eager launches:", same, "/ 65")
    # Round 3's GPU suite saw all 64 replayed rows differ from the eager ones here (whole-suite process only; DESIGN
    # section 4 "graph replay").  decode() now checks every captured step by one replay against the eager step before
    # it uses the graph (chitu_amd.graphs.capture_verified, examined in the failing state by tests/conftest.py), so a
    # difference here means a graph that passed its check at capture and went wrong later.
    ok, err = verdict(rows)
    ok_eager, err_eager = verdict(rows_eager)
    assert same == 65 and err is None and err_eager is None, (
        "graph rows == eager rows", same, "graph run", ok, str(err)[:300], "eager run", ok_eager, str(err_eager)[:300],
        "captures rejected at their replay check:", graphs.unverified_or_retried(),
        "forced launch variants:", _forced_launch_variants(_lib))
    print("HIP vs reference Llama: rel err", ok[0], "greedy agreement", ok[1], "/ 65")
