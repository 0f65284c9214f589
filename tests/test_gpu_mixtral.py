"""Mixtral-family decode step with INT8 W8A8 experts (BASELINE config 4 as a parity-test case)."""

import pytest
import torch

from oracle import llama as ollama
from oracle import mixtral as omix
from tests.util import assert_close, max_rel_to_peak

pytestmark = pytest.mark.gpu


def build(num_hot_req=4):
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.mixtral import MixtralArgs, MixtralDecoder, init_synthetic_

    args = MixtralArgs(dim=1024, n_layers=2, n_heads=8, n_kv_heads=2, vocab_size=2048, ffn_dim=512, num_local_experts=8,
                       num_experts_per_tok=2)
    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=num_hot_req, block_size=256, max_seq_len=1024, device="cuda",
                                n_local_kv_heads=args.n_kv_heads, head_dim=args.head_dim, dtype=torch.bfloat16)
    model = MixtralDecoder(args, cache, HipAttnBackend(local_n_heads=args.n_heads, max_seq_len=1024), max_position_embeddings=1024,
                           device="cuda")
    init_synthetic_(model, seed=0)
    return args, model, cache


def test_router_softmax_topk_renorm():
    from chitu_amd import ops

    g = torch.Generator().manual_seed(1)
    x = torch.randn(19, 1024, generator=g).to(torch.bfloat16)
    gw = (torch.randn(8, 1024, generator=g) * 1024 ** -0.5).to(torch.bfloat16)
    w_ref, i_ref = omix.route(x, gw, 2)
    w, i = ops.gate_deepseek_v3(x.cuda(), gw.cuda(), None, 1, 1, 2, "softmax_renorm", 1.0)
    assert torch.equal(i.cpu(), i_ref)
    assert_close(w.cpu(), w_ref, 1e-2)
    assert torch.allclose(w.float().sum(-1).cpu(), torch.ones(19), atol=1e-2)


def test_router_against_the_reference_block_fixture():
    """The HIP router (softmax -> top-2 -> renormalise) on the inputs of tests/golden/gen_mixtral_router.py vs what the
    reference's SparseMoeBlockHFMixtral.forward produced for them: same experts wherever the reference's second and
    third probabilities are not a near-tie, weights within 1e-2."""
    from chitu_amd import ops
    from tests.util import golden

    g = golden("mixtral_router")
    x = torch.from_numpy(g["x"].copy()).view(torch.bfloat16)
    gate_w = torch.from_numpy(g["gate_w"].copy()).view(torch.bfloat16)
    want = torch.from_numpy(g["weights"].copy()).view(torch.bfloat16).float()
    w, ids = ops.gate_deepseek_v3(x.cuda(), gate_w.cuda(), None, 1, 1, int(g["topk"][0]), "softmax_renorm", 1.0)
    got = torch.zeros_like(want).scatter_(1, ids.cpu(), w.float().cpu())
    probs = torch.softmax(torch.nn.functional.linear(x.float(), gate_w.float()), -1).sort(-1, descending=True).values
    clear = (probs[:, 1] - probs[:, 2]) > 2e-3
    assert clear.sum() >= 60
    assert torch.equal(got[clear] != 0, want[clear] != 0)
    assert (got[clear] - want[clear]).abs().max() < 1e-2


def test_layerwise_parity_graph_replay_and_generate():
    args, model, cache = build()
    params = {k: v.detach().cpu() for k, v in model.named_parameters()}
    bs, reqs = 3, ["r0", "r1", "r2"]
    gen = torch.Generator().manual_seed(7)
    for r, n in zip(reqs, (0, 255, 300)):
        cache.register_sequence(r, n)
        for blk in cache.block_table[r]:
            cache.paged_k_cache[:, blk] = (torch.randn(args.n_layers, 256, 2, 128, generator=gen) * 0.5).to(torch.bfloat16).cuda()
            cache.paged_v_cache[:, blk] = (torch.randn(args.n_layers, 256, 2, 128, generator=gen) * 0.5).to(torch.bfloat16).cuda()
    shadow_k, shadow_v = cache.paged_k_cache.cpu().clone(), cache.paged_v_cache.cpu().clone()
    cache.prepare_cache_decode(reqs)
    cache.prepare_block_table_for_decode(reqs)
    lens = cache.get_gpu_seq_lens_excl_this_decode()[:bs].cpu()
    table = cache.get_gpu_block_table()[:bs].cpu()
    cos, sin = model.cos_table.cpu()[lens.long()], model.sin_table.cpu()[lens.long()]
    x = torch.randn(bs, args.dim, generator=gen).to(torch.bfloat16)
    for i, layer in enumerate(model.layers):
        with torch.inference_mode():
            xm, pend = layer(x.cuda(), None, cos.cuda(), sin.cuda())
        if pend.dim() == 3:  # the un-summed top-2 outputs (the next residual add sums them: moe_sum's arithmetic, one rounding)
            pend = (pend[:, 0].float() + pend[:, 1].float()).to(torch.bfloat16)
        y = (xm + pend).cpu()
        pre = f"layers.{i}."
        y_ref, _, _ = ollama.block(params, pre, x, cos, sin, shadow_k[i], shadow_v[i], table, lens, args.n_heads, 2, 128,
                                   args.norm_eps, rotary="hf-llama",
                                   ffn=lambda hn, pre=pre: omix.sparse_moe(params, pre + "ffn.", hn, 2)[0])
        err = max_rel_to_peak(y, y_ref)
        assert err < 2e-2, (i, err)
        x = y_ref
    tokens = torch.tensor([5, 17, 900], dtype=torch.int64, device="cuda")
    for step in range(2):
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        snap_k, snap_v = cache.paged_k_cache.clone(), cache.paged_v_cache.clone()
        eager = model.decode(tokens, use_graph=False).clone()
        cache.paged_k_cache.copy_(snap_k)
        cache.paged_v_cache.copy_(snap_v)
        graph = model.decode(tokens, use_graph=True).clone()
        assert torch.equal(eager, graph) and torch.isfinite(eager).all()
        tokens = eager.argmax(dim=-1)
        cache.finalize_cache_single_decode(reqs)
    for r in reqs:
        cache.finalize_cache_all_decode(r)
    out1 = model.generate([[3, 4, 5], [7]], 3)
    assert tuple(out1.shape) == (2, 3) and torch.equal(out1, model.generate([[3, 4, 5], [7]], 3))


@pytest.mark.parametrize("bs", [1, 2, 3, 16])
def test_decode_step_with_the_round_6_fusions_equals_the_separate_launches(bs):
    """Routing + moe_align in one launch, attn_norm as the prologue of the qkv GEMM (bs <= 2), the experts' int8 quantisation
    inside ffn_norm, the top-2 sum inside the next residual add (rms_norm(add=<3-D>) or the qkv GEMM's two-term prologue):
    the logits of two decode steps are bit-identical to the step made of the separate launches (mixtral.FUSE = False)."""
    from chitu_amd import mixtral

    args, model, cache = build(num_hot_req=16)
    reqs = [f"f{i}" for i in range(bs)]
    gen = torch.Generator().manual_seed(11 + bs)
    for i, r in enumerate(reqs):
        cache.register_sequence(r, 5 + 13 * i)
    cache.paged_k_cache.copy_((torch.randn(cache.paged_k_cache.shape, generator=gen) * 0.5).to(torch.bfloat16))
    cache.paged_v_cache.copy_((torch.randn(cache.paged_v_cache.shape, generator=gen) * 0.5).to(torch.bfloat16))
    tokens = torch.randint(3, 900, (bs,), generator=gen).cuda()
    outs = {}
    try:
        for fuse in (True, False):
            mixtral.FUSE = fuse
            toks, res = tokens, []
            snap_k, snap_v = cache.paged_k_cache.clone(), cache.paged_v_cache.clone()
            for step in range(2):
                cache.prepare_cache_decode(reqs)
                cache.prepare_block_table_for_decode(reqs)
                lg = model.decode(toks, use_graph=False).clone()
                res.append(lg)
                toks = lg.argmax(dim=-1)
                cache.finalize_cache_single_decode(reqs)
            outs[fuse] = res
            # rewind the two steps for the second pass
            for r in reqs:
                cache.seq_lens[r] -= 2
            cache.paged_k_cache.copy_(snap_k)
            cache.paged_v_cache.copy_(snap_v)
    finally:
        mixtral.FUSE = True
    for a, b in zip(outs[True], outs[False]):
        assert torch.isfinite(a).all() and torch.equal(a, b)
