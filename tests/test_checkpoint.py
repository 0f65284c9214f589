"""chitu_amd/checkpoint.py against the reference's own loader functions (tests/golden/gen_ckpt.py ->
ckpt_preprocess.json: per TP rank the ordered (name, shape, dtype, sha1) list), plus load / save round trips on the
module tree.  Host-only: runs without a GPU."""

import json
import os

import pytest
import torch

from chitu_amd import checkpoint as ck
from tests.util import CKPT_TINY, tensor_digest, tiny_hf_checkpoint

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    with open(os.path.join(HERE, "golden", "ckpt_preprocess.json")) as f:
        return json.load(f)


def _renamed():
    out = {}
    for k, v in tiny_hf_checkpoint().items():
        n = ck.map_hf_name(k)
        if n is not None:
            out[n] = v
    return out


def _args(shard):
    from chitu_amd.deepseek_v3 import DeepSeekV3Args

    keys = ("vocab_size", "dim", "inter_dim", "moe_inter_dim", "n_layers", "n_dense_layers", "n_heads", "n_routed_experts",
            "n_shared_experts", "n_activated_experts", "n_expert_groups", "n_limited_groups", "route_scale", "score_func",
            "q_lora_rank", "kv_lora_rank", "qk_nope_head_dim", "qk_rope_head_dim", "v_head_dim", "rope_theta", "rope_factor")
    return DeepSeekV3Args(**{k: CKPT_TINY[k] for k in keys}, gate_bias=True, shard_degree=shard)


def test_hf_names_map_like_the_reference_loader():
    g = _golden()
    assert sorted(_renamed().keys()) == sorted(g["names_after_rename"])  # (file key order is safetensors', not ours)
    assert ck.map_hf_name("model.layers.61.self_attn.q_a_proj.weight") is None  # MTP layer dropped
    assert ck.map_hf_name("model.layers.3.mlp.gate.e_score_correction_bias") == "layers.3.ffn.gate.bias"
    assert ck.map_hf_name("model.layers.3.mlp.experts.7.down_proj.weight_scale_inv") == "layers.3.ffn.experts.7.w2.scale"
    with pytest.raises(KeyError):
        ck.map_hf_name("model.layers.0.self_attn.rotary_emb.inv_freq")


@pytest.mark.parametrize("rank", [0, 1])
def test_shard_merge_stack_bit_identical_to_the_reference(rank):
    g = _golden()
    mine = ck.preprocess_deepseek_v3(_renamed(), CKPT_TINY["n_routed_experts"], rank, g["tp"])
    got = {k: tensor_digest(v) for k, v in mine.items()}
    want = {r[0]: r[1:] for r in g["ranks"][rank]}
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])


def test_ranks_partition_the_checkpoint():
    """Concatenating the two ranks' shards along the sharded dim gives back the unsharded tensors (size-independent
    property: nothing lost, nothing duplicated)."""
    full = ck.preprocess_deepseek_v3(_renamed(), 4, 0, 1)
    r0 = ck.preprocess_deepseek_v3(_renamed(), 4, 0, 2)
    r1 = ck.preprocess_deepseek_v3(_renamed(), 4, 1, 2)
    for k, t in full.items():
        a, b = r0[k], r1[k]
        if a.shape == t.shape:
            assert torch.equal(a.view(torch.uint8), t.view(torch.uint8)) and torch.equal(b.view(torch.uint8), t.view(torch.uint8))
            continue
        dim = [i for i in range(t.dim()) if a.shape[i] != t.shape[i]]
        assert len(dim) == 1, k
        d = dim[0]
        if k.endswith(("w1w3.weight", "w1w3.scale")):  # merged [w1 shard | w3 shard]: un-merge before comparing
            h, ht = a.shape[d] // 2, t.shape[d] // 2
            lo = torch.cat([a.narrow(d, 0, h), b.narrow(d, 0, h)], dim=d)
            hi = torch.cat([a.narrow(d, h, h), b.narrow(d, h, h)], dim=d)
            assert torch.equal(lo.view(torch.uint8), t.narrow(d, 0, ht).contiguous().view(torch.uint8)), k
            assert torch.equal(hi.view(torch.uint8), t.narrow(d, ht, ht).contiguous().view(torch.uint8)), k
        else:
            assert torch.equal(torch.cat([a, b], dim=d).view(torch.uint8), t.contiguous().view(torch.uint8)), k


def test_uneven_shards_are_refused():
    st = {"layers.0.attn.wq_b.weight": torch.zeros(6, 4)}
    with pytest.raises(ValueError):
        ck.chunk_for_tensor_parallel(st, 0, 4)


@pytest.mark.parametrize("rank", [0, 1])
def test_load_into_module_tree_and_save_round_trip(rank, tmp_path):
    from safetensors.torch import save_file

    from chitu_amd.deepseek_v3 import DeepSeekV3Decoder

    save_file({k: v.contiguous() for k, v in tiny_hf_checkpoint().items()}, str(tmp_path / "model-00001-of-00001.safetensors"))
    model = DeepSeekV3Decoder(_args(2), None, None, max_position_embeddings=64, device="cpu")
    model.layers[1].attn._w_uk_key = "stale"
    ck.load_checkpoint_deepseek_v3(model, str(tmp_path), rank=rank, world=2)
    assert model.layers[1].attn._w_uk_key is None
    want = ck.to_module_names(ck.preprocess_deepseek_v3(_renamed(), 4, rank, 2))
    params = dict(model.named_parameters())
    assert set(params) == set(want)
    for k, p in params.items():
        assert tensor_digest(p) == tensor_digest(want[k]), k
    # spot checks of the layout the kernels assume
    qa = tiny_hf_checkpoint()["model.layers.0.self_attn.q_a_proj.weight"]
    assert torch.equal(params["layers.0.attn.wqkv_a.weight"][:128].view(torch.uint8), qa.view(torch.uint8))
    sh = tiny_hf_checkpoint()["model.layers.1.mlp.shared_experts.up_proj.weight"]
    assert torch.equal(params["layers.1.ffn.w1w3_weight"][4, 128:].view(torch.uint8),
                       torch.chunk(sh, 2, dim=0)[rank].contiguous().view(torch.uint8))  # shared expert = last slot, up half
    # preprocess-and-save, then the skip_preprocess load (script/preprocess_and_save.py)
    out = tmp_path / "pre"
    ck.save_preprocessed(model, str(out), rank)
    model2 = DeepSeekV3Decoder(_args(2), None, None, max_position_embeddings=64, device="cpu")
    ck.load_checkpoint_deepseek_v3(model2, str(out), rank=rank, world=2, skip_preprocess=True)
    for (k, p), (k2, p2) in zip(model.named_parameters(), model2.named_parameters()):
        assert k == k2 and tensor_digest(p) == tensor_digest(p2), k


def test_load_rejects_wrong_shapes_and_fp8_casts():
    from chitu_amd.deepseek_v3 import DeepSeekV3Decoder

    model = DeepSeekV3Decoder(_args(2), None, None, max_position_embeddings=64, device="cpu")
    good = ck.to_module_names(ck.preprocess_deepseek_v3(_renamed(), 4, 0, 2))
    bad = dict(good)
    bad["layers.0.attn.wo.weight"] = good["layers.0.attn.wo.weight"][:, :128]
    with pytest.raises(ValueError):
        ck.load_deepseek_v3(model, bad)
    bad = dict(good)
    bad["layers.0.attn.wo.weight"] = good["layers.0.attn.wo.weight"].to(torch.bfloat16)
    with pytest.raises(TypeError):
        ck.load_deepseek_v3(model, bad)
    bad = dict(good)
    del bad["norm.weight"]
    with pytest.raises(KeyError):
        ck.load_deepseek_v3(model, bad)


def test_q_lora_rank_zero_merges_wq_with_wkv_a():
    st = {"layers.0.attn.wq.weight": torch.arange(12.0).view(6, 2), "layers.0.attn.wkv_a.weight": torch.ones(3, 2),
          "layers.0.attn.wq.scale": torch.zeros(2, 1), "layers.0.attn.wkv_a.scale": torch.ones(1, 1)}
    sharded = ck.chunk_for_tensor_parallel(st, 1, 2)
    out = ck.to_module_names(sharded, q_lora_rank=0)
    assert set(out) == {"layers.0.attn.wq_kv_a.weight", "layers.0.attn.wq_kv_a.scale"}
    assert torch.equal(out["layers.0.attn.wq_kv_a.weight"], torch.cat([st["layers.0.attn.wq.weight"][3:], torch.ones(3, 2)]))


def test_expert_parallel_layout_partitions_the_checkpoint_and_loads():
    """moe_world_size = 2: the routed experts are split by id at full width, the shared expert by
    width; the two ranks together hold every byte of the unsharded layer exactly once, and each rank's dict loads
    strictly into the expert-parallel module tree."""
    from chitu_amd.cache_manager import PagedKVCacheManager
    from chitu_amd.deepseek_v3 import DeepSeekV3Decoder
    from tests import cpu_ops_shim

    nr = CKPT_TINY["n_routed_experts"]
    full = ck.to_module_names(ck.preprocess_deepseek_v3(_renamed(), nr, 0, 1))
    ranks = [ck.to_module_names(ck.preprocess_deepseek_v3(_renamed(), nr, r, 2, moe_world_size=2)) for r in range(2)]
    pre = "layers.1.ffn."
    for part in ("weight", "scale"):
        w13 = full[pre + "w1w3_" + part]  # [nr + 1, 2I(/128), K(/128)], shared last
        w2 = full[pre + "w2_" + part]
        assert torch.equal(torch.cat([r[pre + "w1w3_" + part] for r in ranks]).view(torch.uint8), w13[:nr].view(torch.uint8))
        assert torch.equal(torch.cat([r[pre + "w2_" + part] for r in ranks]).view(torch.uint8), w2[:nr].view(torch.uint8))
        i = w13.shape[1] // 2
        halves = [r[pre + "shared.w1w3." + part] for r in ranks]  # each [gate chunk | up chunk]
        c = halves[0].shape[0] // 2
        gate = torch.cat([h[:c] for h in halves])
        up = torch.cat([h[c:] for h in halves])
        assert torch.equal(gate.view(torch.uint8), w13[nr, :i].view(torch.uint8))
        assert torch.equal(up.view(torch.uint8), w13[nr, i:].view(torch.uint8))
        assert torch.equal(torch.cat([r[pre + "shared.w2." + part] for r in ranks], dim=1).view(torch.uint8), w2[nr].view(torch.uint8))
    # everything that is not an expert is the plain tensor-parallel shard
    tp_rank = [ck.to_module_names(ck.preprocess_deepseek_v3(_renamed(), nr, r, 2)) for r in range(2)]
    for r in range(2):
        for k, t in tp_rank[r].items():
            if ".ffn.w1w3_" in k or ".ffn.w2_" in k:
                continue
            assert torch.equal(ranks[r][k].view(torch.uint8), t.view(torch.uint8)), k
    for r in range(2):
        args = _args(2)
        args.moe_world_size, args.moe_rank = 2, r
        cache = PagedKVCacheManager(0, args.n_layers, num_hot_req=1, block_size=64, max_seq_len=64, device="cpu",
                                    kv_shape_per_sample=(args.kv_lora_rank + args.qk_rope_head_dim,), dtype=torch.bfloat16)
        model = DeepSeekV3Decoder(args, cache, cpu_ops_shim.CpuAttnBackend(args.n_heads // 2), max_position_embeddings=64,
                                  device="cpu")
        ck.load_deepseek_v3(model, ranks[r])  # strict: names, shapes and dtypes all match
        moe = model.layers[1].ffn
        assert moe.expert_map.tolist() == [0, 1, -1, -1] if r == 0 else moe.expert_map.tolist() == [-1, -1, 0, 1]
    with pytest.raises(ValueError):
        ck.preprocess_deepseek_v3(_renamed(), nr, 0, 2, moe_world_size=4)


def test_two_shared_experts_become_two_slots_of_the_routed_width():
    """DeepSeek-V2-Lite layout (n_shared_experts = 2): the HF shared MLP of width 2*I is cut into two expert slots
    of width I -- rows of w1 / w3 inside the merged w1w3, columns of w2, block scales alike -- so that running the
    slots as experts with weight 1 and summing equals the one wide MLP.  n_shared = 1 stays the reference's append."""
    g = torch.Generator().manual_seed(3)
    I, K, nr = 256, 384, 3
    st = {}
    for i in range(nr):
        st[f"layers.1.ffn.experts.{i}.w1.weight"] = torch.randn(I, K, generator=g)
        st[f"layers.1.ffn.experts.{i}.w3.weight"] = torch.randn(I, K, generator=g)
        st[f"layers.1.ffn.experts.{i}.w2.weight"] = torch.randn(K, I, generator=g)
        st[f"layers.1.ffn.experts.{i}.w1.scale"] = torch.rand(I // 128, K // 128, generator=g)
        st[f"layers.1.ffn.experts.{i}.w3.scale"] = torch.rand(I // 128, K // 128, generator=g)
        st[f"layers.1.ffn.experts.{i}.w2.scale"] = torch.rand(K // 128, I // 128, generator=g)
    sw1, sw3, sw2 = torch.randn(2 * I, K, generator=g), torch.randn(2 * I, K, generator=g), torch.randn(K, 2 * I, generator=g)
    st["layers.1.ffn.shared_experts.w1.weight"], st["layers.1.ffn.shared_experts.w3.weight"] = sw1, sw3
    st["layers.1.ffn.shared_experts.w2.weight"] = sw2
    st["layers.1.ffn.shared_experts.w1.scale"] = torch.rand(2 * I // 128, K // 128, generator=g)
    st["layers.1.ffn.shared_experts.w3.scale"] = torch.rand(2 * I // 128, K // 128, generator=g)
    st["layers.1.ffn.shared_experts.w2.scale"] = torch.rand(K // 128, 2 * I // 128, generator=g)
    out = ck.preprocess_deepseek_v3(st, nr, n_shared=2)
    w13, w2 = out["layers.1.ffn.w1w3.weight"], out["layers.1.ffn.w2.weight"]
    assert w13.shape == (nr + 2, 2 * I, K) and w2.shape == (nr + 2, K, I)
    assert out["layers.1.ffn.w1w3.scale"].shape == (nr + 2, 2 * I // 128, K // 128)
    assert out["layers.1.ffn.w2.scale"].shape == (nr + 2, K // 128, I // 128)
    x = torch.randn(5, K, generator=g)
    wide = (torch.nn.functional.silu(x @ sw1.T) * (x @ sw3.T)) @ sw2.T
    slots = sum((torch.nn.functional.silu(x @ w13[nr + j, :I].T) * (x @ w13[nr + j, I:].T)) @ w2[nr + j].T for j in range(2))
    assert torch.allclose(wide, slots, rtol=1e-4, atol=1e-3)
    for j in range(2):
        assert torch.equal(out["layers.1.ffn.w1w3.scale"][nr + j, : I // 128], st["layers.1.ffn.shared_experts.w1.scale"][j * (I // 128):(j + 1) * (I // 128)])
        assert torch.equal(out["layers.1.ffn.w2.scale"][nr + j], st["layers.1.ffn.shared_experts.w2.scale"][:, j * (I // 128):(j + 1) * (I // 128)])
    with pytest.raises(ValueError):  # the wide MLP handed over as ONE slot does not have the routed experts' shape
        ck.preprocess_deepseek_v3(st, nr, n_shared=1)


def test_transposed_w_uk_tracks_the_weights_and_keeps_its_buffer():
    """The derived [H, kv_lora, nope] copy of W_UK is rebuilt IN PLACE when wkv_b changes through a tracked path
    (copy_ / load_state_dict: version counter) and after `refresh_derived_layouts` (writers that go through `.data`),
    so a captured graph keeps reading valid bytes; a model built under inference_mode (no version counters) works."""
    from chitu_amd.deepseek_v3 import DeepSeekV3Decoder, refresh_derived_layouts

    model = DeepSeekV3Decoder(_args(1), None, None, max_position_embeddings=64, device="cpu")
    attn = model.layers[0].attn
    w = attn.wkv_b.weight
    with torch.no_grad():
        w.copy_(torch.randint(0, 120, w.shape, dtype=torch.uint8).view(w.dtype))
    H = attn.n_local_heads

    def expect():
        return w.view(torch.uint8).view(H, 256, attn.kv_lora_rank)[:, :128].transpose(1, 2).contiguous()

    t1 = attn.w_uk_transposed()
    ptr = t1.data_ptr()
    assert torch.equal(t1.view(torch.uint8), expect())
    with torch.no_grad():
        w.copy_(torch.randint(0, 120, w.shape, dtype=torch.uint8).view(w.dtype))  # tracked write
    t2 = attn.w_uk_transposed()
    assert t2.data_ptr() == ptr and torch.equal(t2.view(torch.uint8), expect())
    w.data.view(torch.uint8).fill_(7)  # untracked write
    refresh_derived_layouts(model)
    assert attn.w_uk_transposed().data_ptr() == ptr and torch.equal(attn.w_uk_transposed().view(torch.uint8), expect())
    with torch.inference_mode():
        m2 = DeepSeekV3Decoder(_args(1), None, None, max_position_embeddings=64, device="cpu")
        m2.layers[0].attn.wkv_b.weight.view(torch.uint8).fill_(3)
        assert int(m2.layers[0].attn.w_uk_transposed().view(torch.uint8).max()) == 3


# ---------------------------------------------------------------- Llama family (BASELINE configs 1, 2, 4)
def _llama_golden():
    with open(os.path.join(os.path.dirname(__file__), "golden", "ckpt_preprocess_llama.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("kind", ["llama", "merged", "mixtral"])
@pytest.mark.parametrize("world", [1, 2])
def test_hf_llama_family_shard_and_merge_bit_identical_to_the_reference(kind, world):
    """strip "model." -> (Mixtral renames) -> split merged qkv / gate_up for TP -> TP chunk -> merge per rank: the same
    ordered names, shapes, dtypes and bytes as the reference's own TransformerHFLlama / TransformerHFMixtral loader methods
    (tests/golden/gen_ckpt_llama.py), files with separate projections, files that ship them merged (+ qkv bias), Mixtral."""
    from tests.util import HF_LLAMA_TINY, tiny_hf_llama_checkpoint

    c = HF_LLAMA_TINY
    gold = _llama_golden()[kind][str(world)]
    for rank in range(world):
        st = ck.preprocess_hf_llama(tiny_hf_llama_checkpoint(kind), c["n_heads"], c["n_kv_heads"], c["dim"], rank, world,
                                    mixtral=kind == "mixtral")
        got = [[k] + tensor_digest(v) for k, v in st.items()]
        assert [g[0] for g in got] == [g[0] for g in gold[rank]]
        assert got == gold[rank]


def test_hf_llama_loads_into_the_decoder_and_ranks_partition_it():
    from chitu_amd.llama import LlamaArgs, LlamaDecoder
    from tests.util import HF_LLAMA_TINY, tiny_hf_llama_checkpoint

    c = dict(HF_LLAMA_TINY, dim=512)  # the decoders' attention kernels are built for head_dim 128
    hf = tiny_hf_llama_checkpoint("llama", cfg=c)
    args = LlamaArgs(dim=c["dim"], n_layers=c["n_layers"], n_heads=c["n_heads"], n_kv_heads=c["n_kv_heads"],
                     vocab_size=c["vocab_size"], ffn_dim=c["ffn_dim"])
    model = LlamaDecoder(args, None, None, max_position_embeddings=64, device="cpu")
    st = ck.to_llama_module_names(ck.preprocess_hf_llama(hf, args.n_heads, args.n_kv_heads, args.dim))
    ck.load_deepseek_v3(model, st)  # strict: every parameter named, nothing left over
    p = dict(model.named_parameters())
    a = "model.layers.1.self_attn."
    assert torch.equal(p["layers.1.attn.wqkv"], torch.cat([hf[a + "q_proj.weight"], hf[a + "k_proj.weight"], hf[a + "v_proj.weight"]]))
    assert torch.equal(p["layers.1.ffn.w13"], torch.cat([hf["model.layers.1.mlp.gate_proj.weight"], hf["model.layers.1.mlp.up_proj.weight"]]))
    assert torch.equal(p["head_weight"], hf["lm_head.weight"]) and torch.equal(p["layers.0.attn_norm"], hf["model.layers.0.input_layernorm.weight"])
    # two ranks: heads / FFN width / vocabulary split, every byte of the checkpoint on exactly one rank (norms on both)
    r = [ck.to_llama_module_names(ck.preprocess_hf_llama(hf, args.n_heads, args.n_kv_heads, args.dim, rank, 2)) for rank in range(2)]
    hd, hq, hkv = c["dim"] // c["n_heads"], c["n_heads"], c["n_kv_heads"]
    q = torch.cat([r[0]["layers.0.attn.wqkv"][: hq // 2 * hd], r[1]["layers.0.attn.wqkv"][: hq // 2 * hd]])
    assert torch.equal(q, hf["model.layers.0.self_attn.q_proj.weight"])
    v = torch.cat([r[0]["layers.0.attn.wqkv"][-(hkv // 2) * hd:], r[1]["layers.0.attn.wqkv"][-(hkv // 2) * hd:]])
    assert torch.equal(v, hf["model.layers.0.self_attn.v_proj.weight"])
    assert torch.equal(torch.cat([r[0]["layers.0.attn.wo"], r[1]["layers.0.attn.wo"]], 1), hf["model.layers.0.self_attn.o_proj.weight"])
    assert torch.equal(torch.cat([r[0]["layers.0.ffn.w2"], r[1]["layers.0.ffn.w2"]], 1), hf["model.layers.0.mlp.down_proj.weight"])
    assert torch.equal(torch.cat([r[0]["embed_weight"], r[1]["embed_weight"]]), hf["model.embed_tokens.weight"])
    with pytest.raises(NotImplementedError):  # a qkv bias has no parameter in LlamaDecoder: refused, not dropped
        ck.to_llama_module_names(ck.preprocess_hf_llama(tiny_hf_llama_checkpoint("merged", cfg=c), hq, hkv, c["dim"]))
    with pytest.raises(ValueError):  # 3 ranks do not divide 2 KV heads' rows evenly... nor the vocabulary: refused
        ck.preprocess_hf_llama(hf, hq, hkv, c["dim"], 0, 3)


def test_hf_mixtral_loads_with_int8_experts_quantised_like_simple_w8a8():
    from chitu_amd.mixtral import MixtralArgs, MixtralDecoder
    from chitu_amd.quantize.w8a8 import quant_weight
    from tests.util import HF_LLAMA_TINY, tiny_hf_llama_checkpoint

    c = dict(HF_LLAMA_TINY, dim=512)
    hf = tiny_hf_llama_checkpoint("mixtral", cfg=c)
    args = MixtralArgs(dim=c["dim"], n_layers=c["n_layers"], n_heads=c["n_heads"], n_kv_heads=c["n_kv_heads"], vocab_size=c["vocab_size"],
                       ffn_dim=c["ffn_dim"], num_local_experts=c["num_local_experts"])
    model = MixtralDecoder(args, None, None, max_position_embeddings=64, device="cpu")
    st = ck.preprocess_hf_llama(hf, args.n_heads, args.n_kv_heads, args.dim, mixtral=True, router_row_parallel=False)
    ck.load_deepseek_v3(model, ck.to_mixtral_module_names(st, args.num_local_experts))  # strict
    p = dict(model.named_parameters())
    m = "model.layers.1.block_sparse_moe."
    assert torch.equal(p["layers.1.ffn.gate"], hf[m + "gate.weight"])
    for e in (0, 3):
        w13 = torch.cat([hf[m + f"experts.{e}.w1.weight"], hf[m + f"experts.{e}.w3.weight"]]).to(torch.float16)
        q, s = quant_weight(w13)
        assert torch.equal(p["layers.1.ffn.w13"][e], q) and torch.equal(p["layers.1.ffn.w13_scale"][e], s)
        deq = p["layers.1.ffn.w2"][e].float() * p["layers.1.ffn.w2_scale"][e][:, None]
        ref = hf[m + f"experts.{e}.w2.weight"].float()
        assert (deq - ref).abs().max() <= 0.5 * p["layers.1.ffn.w2_scale"][e].max() * 1.001  # half an int8 step per channel
    # TP = 2 keeps the router whole on both ranks here (the reference shards it along dim and all-reduces the logits)
    r1 = ck.preprocess_hf_llama(hf, args.n_heads, args.n_kv_heads, args.dim, 1, 2, mixtral=True, router_row_parallel=False)
    assert r1["layers.0.mlp.gate.weight"].shape == (c["num_local_experts"], c["dim"])
    assert r1["layers.0.mlp.experts.2.gate_up_proj.weight"].shape == (c["ffn_dim"], c["dim"])  # 2 x ffn / 2 rows


def test_hf_llama_directory_to_decoder_in_one_call(tmp_path):
    from safetensors.torch import save_file

    from chitu_amd.llama import LlamaArgs, LlamaDecoder
    from tests.util import HF_LLAMA_TINY, tiny_hf_llama_checkpoint

    c = dict(HF_LLAMA_TINY, dim=512)
    hf = tiny_hf_llama_checkpoint("llama", cfg=c)
    names = sorted(hf)
    save_file({k: hf[k].contiguous() for k in names[: len(names) // 2]}, str(tmp_path / "model-00001-of-00002.safetensors"))
    save_file({k: hf[k].contiguous() for k in names[len(names) // 2:]}, str(tmp_path / "model-00002-of-00002.safetensors"))
    args = LlamaArgs(dim=c["dim"], n_layers=c["n_layers"], n_heads=c["n_heads"], n_kv_heads=c["n_kv_heads"],
                     vocab_size=c["vocab_size"], ffn_dim=c["ffn_dim"])
    model = LlamaDecoder(args, None, None, max_position_embeddings=64, device="cpu")
    ck.load_checkpoint_hf_llama(model, str(tmp_path))
    p = dict(model.named_parameters())
    assert torch.equal(p["layers.0.attn.wo"], hf["model.layers.0.self_attn.o_proj.weight"])
    assert torch.equal(p["embed_weight"], hf["model.embed_tokens.weight"]) and torch.equal(p["norm"], hf["model.norm.weight"])


def test_meta_format_llama_names_and_shards():
    """TransformerLlama's `consolidated.pth` names -> LlamaDecoder parameters (the map tests/util.ref_llama_fixture uses for
    the reference-run comparisons), TP = 1 and 2."""
    g = torch.Generator().manual_seed(5)
    D, hq, hkv, hd, F, V = 256, 2, 1, 128, 96, 32
    sd = {"tok_embeddings.weight": torch.randn(V, D, generator=g), "norm.weight": torch.randn(D, generator=g),
          "output.weight": torch.randn(V, D, generator=g)}
    for n, shape in (("attention.wq", (hq * hd, D)), ("attention.wk", (hkv * hd, D)), ("attention.wv", (hkv * hd, D)),
                     ("attention.wo", (D, hq * hd)), ("feed_forward.w1", (F, D)), ("feed_forward.w3", (F, D)),
                     ("feed_forward.w2", (D, F))):
        sd[f"layers.0.{n}.weight"] = torch.randn(*shape, generator=g)
    sd["layers.0.attention_norm.weight"] = torch.randn(D, generator=g)
    sd["layers.0.ffn_norm.weight"] = torch.randn(D, generator=g)
    p = ck.preprocess_meta_llama(sd)
    assert sorted(p) == sorted(["embed_weight", "norm", "head_weight", "layers.0.attn.wqkv", "layers.0.attn.wo", "layers.0.ffn.w13",
                                "layers.0.ffn.w2", "layers.0.attn_norm", "layers.0.ffn_norm"])
    assert torch.equal(p["layers.0.attn.wqkv"], torch.cat([sd["layers.0.attention.wq.weight"], sd["layers.0.attention.wk.weight"],
                                                          sd["layers.0.attention.wv.weight"]]))
    assert torch.equal(p["layers.0.ffn.w13"], torch.cat([sd["layers.0.feed_forward.w1.weight"], sd["layers.0.feed_forward.w3.weight"]]))
    sd["layers.0.attention.wk.weight"] = torch.randn(2 * hd, D, generator=g)  # two KV heads so that two ranks divide them
    sd["layers.0.attention.wv.weight"] = torch.randn(2 * hd, D, generator=g)
    r = [ck.preprocess_meta_llama(sd, rank, 2) for rank in range(2)]
    assert r[0]["layers.0.attn.wqkv"].shape == ((hq + 4) // 2 * hd, D) and r[1]["embed_weight"].shape == (V // 2, D)
    assert torch.equal(torch.cat([r[0]["layers.0.attn.wo"], r[1]["layers.0.attn.wo"]], 1), sd["layers.0.attention.wo.weight"])
    assert torch.equal(r[1]["layers.0.attn.wqkv"][:hd], sd["layers.0.attention.wq.weight"][hd:])
    assert torch.equal(r[0]["norm"], sd["norm.weight"]) and torch.equal(r[1]["norm"], sd["norm.weight"])


def test_mixtral_preprocess_once_save_and_reload(tmp_path):
    """script/preprocess_and_save.py's flow for the Llama family: HF files -> this rank's decoder (int8 experts) ->
    model.rank0.safetensors -> a second decoder loaded with skip_preprocess holds the same bytes."""
    from safetensors.torch import save_file

    from chitu_amd.mixtral import MixtralArgs, MixtralDecoder
    from tests.util import HF_LLAMA_TINY, tiny_hf_llama_checkpoint

    c = dict(HF_LLAMA_TINY, dim=512)
    hf = tiny_hf_llama_checkpoint("mixtral", cfg=c)
    os.makedirs(tmp_path / "hf")
    save_file({k: v.contiguous() for k, v in hf.items()}, str(tmp_path / "hf" / "model.safetensors"))
    args = MixtralArgs(dim=c["dim"], n_layers=c["n_layers"], n_heads=c["n_heads"], n_kv_heads=c["n_kv_heads"], vocab_size=c["vocab_size"],
                       ffn_dim=c["ffn_dim"], num_local_experts=c["num_local_experts"])
    a = MixtralDecoder(args, None, None, max_position_embeddings=64, device="cpu")
    ck.load_checkpoint_hf_llama(a, str(tmp_path / "hf"))
    ck.save_preprocessed(a, str(tmp_path / "pre"), rank=0)
    b = MixtralDecoder(args, None, None, max_position_embeddings=64, device="cpu")
    ck.load_checkpoint_hf_llama(b, str(tmp_path / "pre"), skip_preprocess=True)
    for (n, p), (_, q) in zip(a.named_parameters(), b.named_parameters()):
        assert torch.equal(p.view(torch.uint8), q.view(torch.uint8)), n
    with pytest.raises(FileNotFoundError):
        ck.load_checkpoint_hf_llama(b, str(tmp_path / "pre"), rank=1, skip_preprocess=True)
