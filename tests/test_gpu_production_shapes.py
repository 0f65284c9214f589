"""Layer parity at the PRODUCTION shapes of BASELINE.json's configs (the tiny-model tests elsewhere never reach
the launch heuristics these shapes select: expert-GEMM wave / round splits, the one-workgroup routing + sort
kernel, the K-split dense GEMMs, 1024-context MLA splits).

  * config 5: one dense and one MoE decoder layer of DeepSeek-R1 at the per-rank shapes of TP=8 (dim 7168, 16 local
    heads, 256 + 1 experts of width 256, dense FFN width 2304), the model's own router, bs in {1, 16, 32}, ctx 1024;
  * config 3: DeepSeek-V2-Lite (dim 2048, q_lora_rank 0, 64 + 2 experts of width 1408, softmax router), TP=1;
  * config 4: one Mixtral-8x7B layer (dim 4096, 32 / 8 heads, 8 experts of width 14336, INT8 W8A8 experts).
Each HIP layer (chitu_amd.deepseek_v3 / chitu_amd.mixtral) against the CPU oracle (oracle.deepseek.block /
oracle.llama.block + oracle.mixtral.sparse_moe) on identical inputs and the HIP layer's own routing decisions
(the router itself is compared separately: same experts except bf16 near-ties).  Bar: <= 1e-2 of the output's
peak for a whole DeepSeek layer (BASELINE.json north_star's own figure, held per LAYER here although a layer chains ~8
fp8 GEMMs and the attention; measured 4-6e-3) and the appended KV row; the Mixtral INT8 layer keeps 2e-2 (per-token int8
activations: one quantisation step is 2^-7 of the row's peak, twice in a layer; measured 9.6e-3).
"""

import pytest
import torch

from oracle import deepseek as ods
from tests.test_gpu_deepseek import cfg_of
from tests.util import assert_close, max_rel_to_peak

pytestmark = pytest.mark.gpu


def _build_deepseek(args, max_reqs, max_seq, heads):
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Decoder, init_synthetic_

    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=max_reqs, block_size=64, max_seq_len=max_seq, device="cuda",
                                kv_shape_per_sample=(576,), dtype=torch.bfloat16)
    model = DeepSeekV3Decoder(args, cache, HipAttnBackend(local_n_heads=heads, max_seq_len=max_seq),
                              max_position_embeddings=4097, device="cuda")
    init_synthetic_(model, seed=11)
    cache.paged_kv_cache.normal_(0, 0.5, generator=torch.Generator(device="cuda").manual_seed(12))  # seeded: the layer's router
    # is compared on the oracle's own activations, and an unseeded cache made a near-tie flip (a different expert) a matter of luck
    return model, cache


def _deepseek_layers_vs_oracle(args, heads, batches, ctx, bar=1e-2):
    model, cache = _build_deepseek(args, max(batches), ctx + 64, heads)
    cfg = cfg_of(args)
    cfg["H"] = heads  # local heads of this rank
    params = {k: v.detach().cpu() for k, v in model.named_parameters()}
    worst = {}
    for bs in batches:
        reqs = [f"b{bs}_{i}" for i in range(bs)]
        gen = torch.Generator().manual_seed(100 + bs)
        lens = [ctx] + torch.randint(ctx // 2, ctx, (bs - 1,), generator=gen).tolist()  # ragged, the longest first
        for r, n in zip(reqs, lens):
            cache.register_sequence(r, n)
        shadow = cache.paged_kv_cache.cpu().clone()
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        model.prepare_decoding_attn()
        lens_excl = cache.get_gpu_seq_lens_excl_this_decode()[:bs].cpu()
        table = cache.get_gpu_block_table()[:bs].cpu()
        cos, sin = model.cos_table.cpu()[lens_excl.long()], model.sin_table.cpu()[lens_excl.long()]
        x = torch.randn(bs, args.dim, generator=gen).to(torch.bfloat16)
        for i, layer in enumerate(model.layers):
            routing = {}
            if layer.is_moe:
                orig = layer.ffn.gate.forward

                def hooked(inp, _orig=orig, _store=routing, **kw):
                    res = _orig(inp, **kw)
                    k = args.n_activated_experts
                    _store["w"], _store["i"], _store["x"] = res[0][:, :k].cpu(), res[1][:, :k].cpu(), inp.cpu()
                    return res

                layer.ffn.gate.forward = hooked
            with torch.inference_mode():
                xm, pend = layer(x.cuda(), None, cos.cuda(), sin.cuda())
                if pend.dim() == 3:  # un-summed top-k terms: moe_sum's arithmetic
                    pend = pend.float().sum(1).to(torch.bfloat16)
                y = (xm + pend).cpu()
            if layer.is_moe:
                layer.ffn.gate.forward = orig
            rt = (routing["w"], routing["i"]) if layer.is_moe else None
            y_ref, new_cache, _ = ods.block(params, i, x, cos, sin, shadow[i], table, lens_excl, cfg, layer.is_moe, rt)
            assert_close(cache.paged_kv_cache[i].cpu(), new_cache, 1e-2, what=(bs, i, "appended KV row"))
            err = max_rel_to_peak(y, y_ref)
            worst[(bs, i)] = err
            assert err < bar, (bs, i, err)
            if layer.is_moe:
                # the router on the very input the HIP router saw: same experts except bf16 near-ties, and the
                # routing weights of the common experts agree
                w_ref, i_ref = ods.gate(routing["x"], params[f"layers.{i}.ffn.gate.weight"], params.get(f"layers.{i}.ffn.gate.bias"),
                                        cfg["n_groups"], cfg["topk_groups"], cfg["topk"], cfg["score_func"], cfg["route_scale"])
                same = sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(i_ref, rt[1])) / i_ref.numel()
                assert same >= 0.95, (bs, i, same)
            x = y_ref
        for r in reqs:
            cache.finalize_cache_all_decode(r)
    print("production-shape layer parity, worst rel-to-peak per (bs, layer):", {k: round(v, 5) for k, v in worst.items()})
    del model, cache
    torch.cuda.empty_cache()


def test_deepseek_r1_rank_layers_dense_and_moe():
    """BASELINE config 5: layer 0 = dense FFN (width 18432 / 8), layer 1 = MoE (256 routed + 1 shared experts of
    width 2048 / 8, sigmoid router with bias, 8 groups / 4 limited, top-8), 128 / 8 heads, MLA absorb, ctx 1024."""
    from chitu_amd.deepseek_v3 import DeepSeekV3Args

    args = DeepSeekV3Args(shard_degree=8, n_layers=2, n_dense_layers=1)
    _deepseek_layers_vs_oracle(args, heads=16, batches=(1, 16, 32), ctx=1024)


def test_deepseek_v2_lite_layers_dense_and_moe():
    """BASELINE config 3 shapes: dim 2048, 16 heads, q_lora_rank 0, 64 routed + 2 shared experts of width 1408
    (11 K-blocks: the wide-expert path), softmax router, top-6, TP=1."""
    from chitu_amd.deepseek_v3 import DeepSeekV3Args

    args = DeepSeekV3Args(vocab_size=102400, dim=2048, inter_dim=11008, moe_inter_dim=1408, n_layers=2, n_dense_layers=1,
                          n_heads=16, n_routed_experts=64, n_shared_experts=2, n_activated_experts=6, n_expert_groups=1,
                          n_limited_groups=1, route_scale=1.0, score_func="softmax", q_lora_rank=0, gate_bias=False,
                          shard_degree=1)
    _deepseek_layers_vs_oracle(args, heads=16, batches=(1, 16), ctx=1024)


def test_mixtral_8x7b_layer_int8():
    """BASELINE config 4 shapes: one Mixtral-8x7B decoder layer (dim 4096, 32 query / 8 KV heads of 128, 8 experts of
    width 14336 as INT8 W8A8, top-2 softmax-renormalised router), bs in {1, 16}, ctx 1024, page 256."""
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.mixtral import MixtralArgs, MixtralDecoder, init_synthetic_
    from oracle import llama as ollama
    from oracle import mixtral as omix

    args = MixtralArgs(n_layers=1)
    ctx = 1024
    cache = PagedKVCacheManager(0, 1, num_hot_req=16, block_size=256, max_seq_len=ctx + 256, device="cuda",
                                n_local_kv_heads=args.n_kv_heads, head_dim=args.head_dim, dtype=torch.bfloat16)
    model = MixtralDecoder(args, cache, HipAttnBackend(local_n_heads=args.n_heads, max_seq_len=ctx + 256),
                           max_position_embeddings=ctx + 256, device="cuda")
    init_synthetic_(model, seed=4)
    kv_gen = torch.Generator(device="cuda").manual_seed(13)  # seeded (see _build_deepseek): one run of this test in three
    # landed on a router near-tie with unseeded caches (0.30 instead of 0.007)
    cache.paged_k_cache.normal_(0, 0.5, generator=kv_gen)
    cache.paged_v_cache.normal_(0, 0.5, generator=kv_gen)
    params = {k: v.detach().cpu() for k, v in model.named_parameters()}
    for bs in (1, 16):
        reqs = [f"m{bs}_{i}" for i in range(bs)]
        gen = torch.Generator().manual_seed(200 + bs)
        lens_in = [ctx] + torch.randint(ctx // 2, ctx, (bs - 1,), generator=gen).tolist()
        for r, n in zip(reqs, lens_in):
            cache.register_sequence(r, n)
        shadow_k, shadow_v = cache.paged_k_cache.cpu().clone(), cache.paged_v_cache.cpu().clone()
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        lens = cache.get_gpu_seq_lens_excl_this_decode()[:bs].cpu()
        table = cache.get_gpu_block_table()[:bs].cpu()
        cos, sin = model.cos_table.cpu()[lens.long()], model.sin_table.cpu()[lens.long()]
        x = torch.randn(bs, args.dim, generator=gen).to(torch.bfloat16)
        with torch.inference_mode():
            xm, pend = model.layers[0](x.cuda(), None, cos.cuda(), sin.cuda())
        if pend.dim() == 3:  # the un-summed top-2 expert outputs (round 6): the next residual add sums them, one rounding
            pend = pend.float().sum(1).to(torch.bfloat16)
        y = (xm + pend).cpu()
        y_ref, _, _ = ollama.block(params, "layers.0.", x, cos, sin, shadow_k[0], shadow_v[0], table, lens, args.n_heads,
                                   args.n_kv_heads, args.head_dim, args.norm_eps, rotary="hf-llama",
                                   ffn=lambda hn: omix.sparse_moe(params, "layers.0.ffn.", hn, args.num_experts_per_tok)[0])
        err = max_rel_to_peak(y, y_ref)
        print(f"mixtral-8x7b layer bs={bs}: rel-to-peak {err:.5f}")
        assert err < 2e-2, (bs, err)
        for r in reqs:
            cache.finalize_cache_all_decode(r)
