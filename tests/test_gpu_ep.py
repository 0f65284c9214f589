"""Expert parallelism on the GPU: one rank's MoE partial sum vs the oracle, the ranks' sum vs
the unsharded fused layer, and a full expert-parallel rank's decode step under hipGraph replay.  The
collective wiring itself (all-reduce as the combine) is covered by tests/test_tp_gloo.py on CPU."""

import copy

import pytest
import torch

from oracle import deepseek as ods
from tests.test_gpu_deepseek import cfg_of, tiny_args, v2lite_like_args
from tests.test_tp_gloo import _shard_ep
from tests.util import assert_close, max_rel_to_peak

pytestmark = pytest.mark.gpu


def _full_moe(args, seed=3):
    from chitu_amd.deepseek_v3 import MoEDeepSeekV3, init_synthetic_

    a = copy.copy(args)
    a.shard_degree = 1
    return init_synthetic_(MoEDeepSeekV3(a, device="cuda"), seed=seed)


def _ep_moe(args, full_sd, rank, ep):
    from chitu_amd.deepseek_v3 import MoEDeepSeekV3

    a = copy.copy(args)
    a.shard_degree, a.moe_world_size, a.moe_rank = ep, ep, rank
    m = MoEDeepSeekV3(a, device="cuda")
    sd = _shard_ep({"layers.1.ffn." + k: v for k, v in full_sd.items()}, args, rank, ep)
    sd = {k[len("layers.1.ffn."):]: v for k, v in sd.items()}
    assert set(sd) == {k for k, _ in m.named_parameters()}
    for k, p in m.named_parameters():
        assert p.shape == sd[k].shape, (k, p.shape, sd[k].shape)
        p.data.copy_(sd[k])
    return m


@pytest.mark.parametrize("make_args,ep", [(tiny_args, 2), (tiny_args, 4), (v2lite_like_args, 2)],
                         ids=["v3_like-ep2", "v3_like-ep4", "v2lite_like-ep2"])
@pytest.mark.parametrize("bs", [1, 19])
def test_expert_parallel_ranks_match_the_oracle_and_sum_to_the_full_layer(make_args, ep, bs):
    from chitu_amd import ops

    args = make_args()
    if args.moe_inter_dim * args.n_shared_experts % (ep * 128):
        args.moe_inter_dim = 512  # the shared width per rank must keep whole 128-blocks
    cfg = cfg_of(args)
    full = _full_moe(args)
    full_sd = {k: v.detach().cpu() for k, v in full.named_parameters()}
    g = torch.Generator().manual_seed(bs + ep)
    x = torch.randn(bs, args.dim, generator=g).to(torch.bfloat16).cuda()
    norm_w = torch.ones(args.dim, dtype=torch.bfloat16, device="cuda")
    hn, hq, hs = ops.rms_norm(x, norm_w, 1e-6, out_bf16=True, quant="group")
    hn_cpu = hn.cpu()
    with torch.inference_mode():
        w, idx = full.gate(hn)
        y_full = full(hn.clone(), (hq, hs)).float().cpu()
    routing = (w.cpu(), idx.cpu())
    total = torch.zeros(bs, args.dim)
    hit_any_remote = False
    for r in range(ep):
        m = _ep_moe(args, full_sd, r, ep)
        assert m.expert_map.dtype == torch.int32 and int((m.expert_map >= 0).sum()) == args.n_routed_experts // ep
        with torch.inference_mode():
            y = m(hn.clone(), (hq, hs))
            y2 = m(hn.clone(), (hq, hs))
        assert torch.equal(y, y2)  # deterministic
        p = {k: v.detach().cpu() for k, v in m.named_parameters()}
        y_ref, _ = ods.moe_layer_ep(p, "", hn_cpu, cfg, m.expert_map.cpu(), routing=routing)
        assert_close(y.cpu(), y_ref, 1e-2, atol_frac=1.0, what=r)  # a rank's partial sum: few terms, near-zero elements carry their neighbours' rounding
        hit_any_remote |= bool((m.expert_map.cpu()[routing[1]] < 0).any())
        total += y.float().cpu()
    assert hit_any_remote
    # the all-reduce's arithmetic (bf16 partial sums per rank) vs the unsharded fused layer
    assert_close(total.to(torch.bfloat16), y_full.to(torch.bfloat16), 2e-2)


def test_expert_parallel_rank_decode_step_graph_equals_eager():
    from chitu_amd.attn_backend import HipAttnBackend
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Decoder, init_synthetic_

    args = tiny_args()
    args.shard_degree, args.moe_world_size, args.moe_rank = 2, 2, 1
    cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=4, block_size=64, max_seq_len=512, device="cuda",
                                kv_shape_per_sample=(576,), dtype=torch.bfloat16)
    be = HipAttnBackend(local_n_heads=args.n_heads // 2, max_seq_len=512)
    model = DeepSeekV3Decoder(args, cache, be, max_position_embeddings=512, device="cuda")
    init_synthetic_(model, seed=0)
    moe = model.layers[1].ffn
    assert moe.w1w3_weight.shape == (8, 2 * args.moe_inter_dim, args.dim) and moe.shared.inter == args.moe_inter_dim // 2
    reqs = ["a", "b", "c"]
    gen = torch.Generator().manual_seed(1)
    for r, n in zip(reqs, (3, 64, 200)):
        cache.register_sequence(r, n)
        for blk in cache.block_table[r]:
            cache.paged_kv_cache[:, blk] = (torch.randn(args.n_layers, 64, 576, generator=gen) * 0.5).to(torch.bfloat16).cuda()
    tokens = torch.tensor([5, 17, 300], dtype=torch.int64, device="cuda")
    for _ in range(3):
        cache.prepare_cache_decode(reqs)
        cache.prepare_block_table_for_decode(reqs)
        snap = cache.paged_kv_cache.clone()
        eager = model.decode(tokens, use_graph=False).clone()
        cache.paged_kv_cache.copy_(snap)
        graph = model.decode(tokens, use_graph=True).clone()
        assert torch.equal(eager, graph) and torch.isfinite(eager).all()
        tokens = eager.argmax(dim=-1)
        cache.finalize_cache_single_decode(reqs)
