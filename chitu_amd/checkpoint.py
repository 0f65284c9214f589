"""Checkpoint preprocessing for the DeepSeek-V3 / R1 decode path: Hugging Face names -> this
package's module tree, tensor-parallel shards, merged / stacked FP8 tensors, per-rank preprocessed files -- and, at the
end of the file, the same for the Llama family (HF Llama / Mixtral files -> LlamaDecoder / MixtralDecoder: configs 1, 2, 4).

Restates, as one table-driven pipeline, what the reference spreads over
  * chitu/backend.py:431-481        load_state_dict_deepseek_v3  (HF -> chitu names, MTP layer 61 dropped)
  * chitu/models/model.py:332-370   _chunk_checkpoint_for_tensor_parallel  (column: dim 0, row: dim 1; the FP8
                                    block scales are chunked like their weights)
  * chitu/models/model_deepseek_v3.py:1167-1288  merge wq_a|wkv_a -> wqkv_a, w1|w3 -> w1w3, stack the routed experts
                                    with the shared expert as the last slot
  * script/preprocess_and_save.py   one `model.rank{r}.safetensors` per rank, loaded with skip_preprocess
in the reference's order: rename -> TP chunk -> merges -> stack.  `preprocess_deepseek_v3` returns tensors under the
REFERENCE'S final per-rank names (that is what tests/golden/ckpt_preprocess.json pins, digest by digest, against the
reference's own functions); `to_module_names` then maps them onto chitu_amd.deepseek_v3's parameters
(embed.weight -> embed_weight, ffn.w1w3.weight [E,..] -> ffn.w1w3_weight, ...).

Host-only code: nothing here touches the GPU; tensors stay views of the loaded checkpoint wherever the reference's
are (chunks), and are materialised once by the merges.
"""

import glob
import os
import re
from typing import Dict, Iterable, Mapping, Optional

import torch

BLOCK = 128

# last-but-one dotted component of a HF name -> chitu component (backend.py:451-470)
_HF_COMPONENT = {
    "embed_tokens": "embed",
    "input_layernorm": "attn_norm",
    "post_attention_layernorm": "ffn_norm",
    "q_proj": "wq",
    "q_a_proj": "wq_a",
    "q_a_layernorm": "q_norm",
    "q_b_proj": "wq_b",
    "kv_a_proj_with_mqa": "wkv_a",
    "kv_a_layernorm": "kv_norm",
    "kv_b_proj": "wkv_b",
    "o_proj": "wo",
    "gate": "gate",
    "gate_proj": "w1",
    "down_proj": "w2",
    "up_proj": "w3",
    "norm": "norm",
    "lm_head": "head",
    "scale": "scale",
}
_HF_SUBSTRINGS = (("self_attn", "attn"), ("mlp", "ffn"), ("weight_scale_inv", "scale"), ("e_score_correction_bias", "bias"))

# model_deepseek_v3.py:1147-1153
COLUMN_PARALLEL = ("embed", "wq_b", "wkv_b", "w1", "w3", "head", "wq")
ROW_PARALLEL = ("wo", "w2")


def map_hf_name(name: str, n_layers: int = 61) -> Optional[str]:
    """HF parameter name -> chitu name, or None for the multi-token-prediction layers the decode path never runs
    (`model.layers.<i>` with i >= n_layers; backend.py:443-444 hard-codes the one of V3/R1, 61)."""
    m = re.search(r"(?:^|\.)layers\.(\d+)\.", name)
    if m and int(m.group(1)) >= n_layers:
        return None
    if name.startswith("model."):
        name = name[len("model."):]
    for old, new in _HF_SUBSTRINGS:
        name = name.replace(old, new)
    parts = name.split(".")
    comp = parts[-2]
    if comp not in _HF_COMPONENT:
        raise KeyError(f"unknown checkpoint component '{comp}' in '{name}'")
    # the reference does name.replace(key, new_key) on the whole string; the component names never collide with
    # other substrings of a DeepSeek-V3 name except inside themselves, so replacing the component is the same
    return name.replace(comp, _HF_COMPONENT[comp])


def read_safetensors_dir(path: str, skip_preprocess: bool = False, rank: int = 0, n_layers: int = 61) -> Dict[str, torch.Tensor]:
    """All tensors of `path/*.safetensors` under chitu names (or, with skip_preprocess, this rank's
    `model.rank{rank}.safetensors` under the names it was saved with; backend.py:431-481)."""
    from safetensors import safe_open

    pattern = os.path.join(path, f"model.rank{rank}.safetensors" if skip_preprocess else "*.safetensors")
    files = sorted(glob.glob(pattern))
    if not files:
        raise FileNotFoundError(pattern)
    out = {}
    for fp in files:
        if not skip_preprocess and re.search(r"model\.rank\d+\.safetensors$", fp):
            continue
        with safe_open(fp, framework="pt", device="cpu") as f:
            for name in f.keys():
                new = name if skip_preprocess else map_hf_name(name, n_layers)
                if new is not None:
                    out[new] = f.get_tensor(name)
    return out


def _is_layer(layer: str, full: str) -> bool:  # chitu/utils.py:34-39
    return f".{layer}." in full or full.startswith(layer + ".") or full.endswith("." + layer)


def _chunk(t: torch.Tensor, world: int, rank: int, dim: int, name: str) -> torch.Tensor:
    if t.shape[dim] % world:
        # torch.chunk would hand out ragged pieces (and fewer than `world` of them): the reference silently
        # mis-shards there; refuse instead
        raise ValueError(f"{name}: dim {dim} of {tuple(t.shape)} does not divide by tp={world}")
    return torch.chunk(t, world, dim=dim)[rank]


def chunk_for_tensor_parallel(state: Mapping[str, torch.Tensor], rank: int, world: int,
                              column: Iterable[str] = COLUMN_PARALLEL, row: Iterable[str] = ROW_PARALLEL,
                              keep_whole: str = "") -> Dict[str, torch.Tensor]:
    """models/model.py:332-370 for type deepseek-v3: weights AND block scales of column-parallel layers are chunked
    along dim 0, of row-parallel layers along dim 1; column biases along the last dim; row biases stay on rank 0;
    everything else (norms, wq_a / wkv_a, router) is replicated."""
    if world == 1:
        return dict(state)
    out = {}
    for name, t in state.items():
        kind = name.rsplit(".", 1)[-1]
        if keep_whole and keep_whole in name:  # expert parallelism: the routed experts are partitioned by id instead
            out[name] = t
        elif any(_is_layer(s, name) for s in column):
            if kind in ("weight", "scale"):
                out[name] = _chunk(t, world, rank, 0, name)
            elif kind == "bias":
                out[name] = _chunk(t, world, rank, -1, name)
            else:
                raise ValueError(f"illegal parallel tensor {name}")
        elif any(_is_layer(s, name) for s in row):
            if kind in ("weight", "scale"):
                out[name] = _chunk(t, world, rank, 1, name)
            elif kind == "bias":
                if rank == 0:
                    out[name] = t
            else:
                raise ValueError(f"illegal parallel tensor {name}")
        else:
            out[name] = t
    return out


def _merge_pairs(state: Mapping[str, torch.Tensor], first: str, second: str, merged: str) -> Dict[str, torch.Tensor]:
    """`<p>.first.<part>` + `<p>.second.<part>` -> `<p>.merged.<part>` = cat along dim 0, for part in weight / scale /
    bias."""
    out = {}
    for k, t in state.items():
        head, _, part = k.rpartition(".")
        if head.endswith("." + first) and part in ("weight", "scale", "bias"):
            prefix = head[: -len(first)]
            other = prefix + second + "." + part
            if other not in state:
                raise KeyError(f"{k} has no partner {other}")
            if prefix + merged + "." + part in state:
                raise KeyError(f"{prefix + merged}.{part} already present")
            out[prefix + merged + "." + part] = torch.cat([t, state[other]], dim=0)
        elif head.endswith("." + second) and part in ("weight", "scale", "bias"):
            continue
        else:
            out[k] = t
    return out


def merge_qkv(state):
    """wq_a | wkv_a -> wqkv_a (model_deepseek_v3.py:1197-1231).  Block scales concatenate row-wise, which is
    block-aligned because q_lora_rank % 128 == 0."""
    return _merge_pairs(state, "wq_a", "wkv_a", "wqkv_a")


def merge_gate_up(state):
    """w1 | w3 -> w1w3 for the dense MLP, every routed expert and the shared expert (:1233-1271)."""
    return _merge_pairs(state, "w1", "w3", "w1w3")


def _split_shared(t: torch.Tensor, w: str, part: str, n_shared: int):
    """The HF shared MLP is ONE MLP of width n_shared * I (DeepSeek-V2-Lite: 2 x 1408); the fused path runs it
    as n_shared expert slots of width I.  Slice j of the width: rows of w1 / w3 (each half of a merged w1w3),
    columns of w2; block scales likewise (I % 128 == 0 keeps 128-blocks whole)."""
    if n_shared == 1:
        return [t]
    if part == "bias":
        raise ValueError("a shared-expert bias cannot be split over expert slots")
    if w == "w2":
        if t.shape[1] % n_shared:
            raise ValueError(f"shared_experts.w2.{part}: width {t.shape[1]} not divisible by n_shared_experts={n_shared}")
        return list(t.chunk(n_shared, dim=1))
    halves = t.chunk(2, dim=0) if w == "w1w3" else (t,)
    if any(h.shape[0] % n_shared for h in halves):
        raise ValueError(f"shared_experts.{w}.{part}: width {halves[0].shape[0]} not divisible by n_shared_experts={n_shared}")
    pieces = [h.chunk(n_shared, dim=0) for h in halves]
    return [torch.cat([pc[j] for pc in pieces], dim=0) for j in range(n_shared)]


def merge_experts(state: Mapping[str, torch.Tensor], n_routed: int, n_shared: int = 1) -> Dict[str, torch.Tensor]:
    """`<p>.experts.{i}.<w>.<part>` (i < n_routed) + `<p>.shared_experts.<w>.<part>` -> `<p>.<w>.<part>` stacked
    [n_routed + n_shared, ...], shared experts last (:1167-1195; the reference appends the shared MLP as ONE slot,
    i.e. n_shared == 1 -- V3 / R1).  n_shared > 1 (DeepSeek-V2-Lite): the shared MLP's width is cut into n_shared
    slots of the routed experts' width, the layout MoEDeepSeekV3 runs."""
    out = {}
    pat = re.compile(r"^(.*\.)experts\.0\.(w1w3|w1|w2|w3)\.(weight|scale|bias)$")
    for k, t in state.items():
        m = pat.match(k)
        if m:
            prefix, w, part = m.groups()
            parts = [state[f"{prefix}experts.{i}.{w}.{part}"] for i in range(n_routed)]
            shared = _split_shared(state[f"{prefix}shared_experts.{w}.{part}"], w, part, n_shared)
            if any(sh.shape != parts[0].shape for sh in shared):
                raise ValueError(f"{prefix}shared_experts.{w}.{part}: slot shape {tuple(shared[0].shape)} != routed expert "
                                 f"{tuple(parts[0].shape)} (n_shared_experts={n_shared})")
            out[f"{prefix}{w}.{part}"] = torch.stack(parts + shared, dim=0)
        elif ".experts." in k or ".shared_experts." in k:
            continue
        else:
            out[k] = t
    return out


def select_local_experts(state: Mapping[str, torch.Tensor], n_routed: int, moe_rank: int, moe_world_size: int) -> Dict[str, torch.Tensor]:
    """Expert parallelism (the reference's hooks: model_deepseek_v3.py:870-880): keep routed experts
    [moe_rank * n_local, (moe_rank + 1) * n_local) and renumber them from 0 -- the order `expert_map` assigns local
    ids in; everything else is passed through."""
    if n_routed % moe_world_size:
        raise ValueError(f"Number of experts must be divisible by world size (world_size={moe_world_size})")
    n_local = n_routed // moe_world_size
    lo = moe_rank * n_local
    pat = re.compile(r"^(.*\.experts\.)(\d+)(\..*)$")
    out = {}
    for k, t in state.items():
        m = pat.match(k)
        if not m:
            out[k] = t
            continue
        i = int(m.group(2))
        if lo <= i < lo + n_local:
            out[f"{m.group(1)}{i - lo}{m.group(3)}"] = t
    return out


def merge_experts_expert_parallel(state: Mapping[str, torch.Tensor], n_local: int) -> Dict[str, torch.Tensor]:
    """The expert-parallel counterpart of `merge_experts`: local routed experts stacked `[n_local, ...]` under
    `<p>.<w>.<part>`; the (already width-chunked) shared experts stay a plain MLP under `<p>.shared.<w>.<part>`
    (chitu_amd.deepseek_v3.MoEDeepSeekV3 with moe_world_size > 1)."""
    out = {}
    pat = re.compile(r"^(.*\.)experts\.0\.(w1w3|w1|w2|w3)\.(weight|scale|bias)$")
    for k, t in state.items():
        m = pat.match(k)
        if m:
            prefix, w, part = m.groups()
            out[f"{prefix}{w}.{part}"] = torch.stack([state[f"{prefix}experts.{i}.{w}.{part}"] for i in range(n_local)], dim=0)
        elif ".experts." in k:
            continue
        elif ".shared_experts." in k:
            out[k.replace(".shared_experts.", ".shared.")] = t
        else:
            out[k] = t
    return out


def preprocess_deepseek_v3(state: Mapping[str, torch.Tensor], n_routed: int, rank: int = 0, world: int = 1,
                           merge_qkv_gate_up: bool = True, moe_world_size: int = 1, n_shared: int = 1) -> Dict[str, torch.Tensor]:
    """chitu-named full checkpoint -> this rank's tensors under the reference's final names
    (load_state_dict_parallel, models/model.py:372-390 + TransformerDeepSeekV3.load_state_dict :1273-1288).

    moe_world_size > 1 (== world): the expert-parallel layout -- attention, dense MLPs, embeddings and the SHARED
    experts are tensor-parallel chunks as before, the routed experts are partitioned by id and kept at full width."""
    if moe_world_size > 1:
        if moe_world_size != world:
            raise ValueError("experts are partitioned over the tensor-parallel ranks: moe_world_size must equal world")
        if not merge_qkv_gate_up:
            raise ValueError("the expert-parallel module tree uses the merged w1w3 layout")
        state = select_local_experts(state, n_routed, rank, moe_world_size)
        state = chunk_for_tensor_parallel(state, rank, world, keep_whole=".experts.")
        state = merge_gate_up(merge_qkv(state))
        return merge_experts_expert_parallel(state, n_routed // moe_world_size)
    state = chunk_for_tensor_parallel(state, rank, world)
    if merge_qkv_gate_up:
        state = merge_gate_up(merge_qkv(state))
    return merge_experts(state, n_routed, n_shared)


def to_module_names(state: Mapping[str, torch.Tensor], q_lora_rank: int = 1) -> Dict[str, torch.Tensor]:
    """Reference per-rank names -> chitu_amd.deepseek_v3 parameter names.

    embed.weight / head.weight are plain parameters here (embed_weight / head_weight); the stacked experts are
    parameters of the MoE module (ffn.w1w3_weight ...), the dense MLP keeps module names; with q_lora_rank == 0
    (DeepSeek-V2-Lite, SURVEY gap G1) wq | wkv_a become the merged wq_kv_a GEMM."""
    out = {}
    for k, t in state.items():
        if k in ("embed.weight", "head.weight"):
            out[k.replace(".", "_")] = t
            continue
        m = re.match(r"^(.*\.ffn\.)(w1w3|w2)\.(weight|scale)$", k)
        if m and t.dim() == 3:
            out[f"{m.group(1)}{m.group(2)}_{m.group(3)}"] = t
            continue
        out[k] = t
    if q_lora_rank == 0:
        out = _merge_pairs(out, "wq", "wkv_a", "wq_kv_a")
    return out


def load_deepseek_v3(model: torch.nn.Module, state: Mapping[str, torch.Tensor], strict: bool = True) -> None:
    """Copy module-named tensors into `model`'s parameters in place (addresses captured by hipGraphs stay valid),
    with shape / dtype checks instead of nn.Module.load_state_dict's silent fp8 casts.  Derived layouts cached by
    the modules (the transposed W_UK of the absorbed MLA) are marked stale so they are rebuilt from the new weights."""
    params = dict(model.named_parameters())
    missing = sorted(set(params) - set(state))
    unexpected = sorted(set(state) - set(params))
    if strict and (missing or unexpected):
        raise KeyError(f"checkpoint does not match the model: missing {missing[:8]}{'...' if len(missing) > 8 else ''}, "
                       f"unexpected {unexpected[:8]}{'...' if len(unexpected) > 8 else ''}")
    with torch.no_grad():
        for name, p in params.items():
            if name not in state:
                continue
            t = state[name]
            if tuple(t.shape) != tuple(p.shape):
                raise ValueError(f"{name}: checkpoint {tuple(t.shape)} vs model {tuple(p.shape)}")
            if t.dtype != p.dtype:
                if p.element_size() == 1 or t.element_size() == 1:
                    raise TypeError(f"{name}: checkpoint {t.dtype} vs model {p.dtype} (no implicit fp8 conversion)")
                t = t.to(p.dtype)
            p.copy_(t)
    for mod in model.modules():
        if hasattr(mod, "_w_uk_key"):
            mod._w_uk_key = None  # rebuilt in place on next use: buffers captured by hipGraphs stay valid


def load_checkpoint_deepseek_v3(model, path: str, rank: int = 0, world: int = 1, skip_preprocess: bool = False) -> None:
    """HF directory (or a directory written by `save_preprocessed`) -> `model`, one call."""
    args = model.args
    state = read_safetensors_dir(path, skip_preprocess=skip_preprocess, rank=rank, n_layers=args.n_layers)
    if not skip_preprocess:
        state = to_module_names(preprocess_deepseek_v3(state, args.n_routed_experts, rank, world,
                                                       moe_world_size=args.moe_world_size), args.q_lora_rank)
    load_deepseek_v3(model, state)


def save_preprocessed(model: torch.nn.Module, target_dir: str, rank: int = 0) -> str:
    """script/preprocess_and_save.py: this rank's parameters, already sharded / merged / stacked, as
    `model.rank{rank}.safetensors`; `load_checkpoint_deepseek_v3(..., skip_preprocess=True)` reads it back."""
    from safetensors.torch import save_file

    os.makedirs(target_dir, exist_ok=True)
    fp = os.path.join(target_dir, f"model.rank{rank}.safetensors")
    save_file({k: v.detach().cpu().contiguous() for k, v in model.named_parameters()}, fp)
    return fp


# ---------------------------------------------------------------- Llama family (BASELINE configs 1, 2 and 4)
# The reference's HF-Llama / HF-Mixtral load path, restated like the DeepSeek one above:
#   chitu/backend.py:374-380                    the "model." prefix of HF names is dropped
#   chitu/models/model_hf_mixtral.py:171-178    Mixtral: block_sparse_moe -> mlp, w1 / w3 / w2 -> gate / up / down_proj
#   chitu/models/model_hf_llama.py:595-600      tensor parallelism: files that ship qkv / gate_up MERGED are split first
#   chitu/models/model.py:332-370               column-parallel tensors chunked along dim 0 (biases: last dim), row-parallel
#                                               along dim 1 (biases on rank 0 only)
#   chitu/models/model_hf_llama.py:506-566      q | k | v -> qkv_proj, gate | up -> gate_up_proj, per rank
# `preprocess_hf_llama` returns tensors under the REFERENCE'S final per-rank names (pinned digest by digest against the
# reference's own methods, tests/golden/gen_ckpt_llama.py); `to_llama_module_names` / `to_mixtral_module_names` map them
# onto chitu_amd.llama.LlamaDecoder / chitu_amd.mixtral.MixtralDecoder.

HF_LLAMA_COLUMN = ("qkv_proj", "q_proj", "k_proj", "v_proj", "gate_up_proj", "gate_proj", "up_proj", "lm_head", "embed_tokens")
HF_LLAMA_ROW = ("down_proj", "o_proj")  # model_hf_llama.py:403-417; Mixtral adds its router "gate" (model_hf_mixtral.py:157-160)


def strip_model_prefix(state: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {(k[len("model."):] if k.startswith("model.") else k): v for k, v in state.items()}


def map_mixtral_names(state: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in state.items():
        for a, b in ((".block_sparse_moe.", ".mlp."), (".w1.", ".gate_proj."), (".w3.", ".up_proj."), (".w2.", ".down_proj.")):
            k = k.replace(a, b)
        out[k] = v
    return out


def _split_merged(state: Mapping[str, torch.Tensor], merged: str, parts, sizes_of) -> Dict[str, torch.Tensor]:
    """`<p>.merged.{weight,bias}` -> `<p>.part.{weight,bias}` for part in parts, split along dim 0 into sizes_of(tensor)."""
    out = {}
    for k, t in state.items():
        head, _, kind = k.rpartition(".")
        if head.endswith("." + merged) and kind in ("weight", "bias"):
            prefix = head[: -len(merged)]
            for part in parts:
                if prefix + part + "." + kind in state:
                    raise KeyError(f"{k}: {prefix + part}.{kind} is present as well")
            for part, piece in zip(parts, t.split(sizes_of(t), dim=0)):
                out[prefix + part + "." + kind] = piece
        else:
            out[k] = t
    return out


def split_merged_qkv(state, n_heads: int, n_kv_heads: int, head_dim: int):
    """qkv_proj -> q_proj | k_proj | v_proj (model_hf_llama.py:428-481): needed before TP sharding, where every part is
    cut by heads."""
    sizes = [n_heads * head_dim, n_kv_heads * head_dim, n_kv_heads * head_dim]
    return _split_merged(state, "qkv_proj", ("q_proj", "k_proj", "v_proj"), lambda t: sizes)


def split_merged_gate_up(state):
    """gate_up_proj -> gate_proj | up_proj, two equal halves (model_hf_llama.py:483-504)."""
    return _split_merged(state, "gate_up_proj", ("gate_proj", "up_proj"), lambda t: [t.shape[0] // 2, t.shape[0] - t.shape[0] // 2])


def _merge_group(state: Mapping[str, torch.Tensor], parts, merged: str) -> Dict[str, torch.Tensor]:
    """`<p>.parts[0].<kind>` .. `<p>.parts[-1].<kind>` -> `<p>.merged.<kind>` = cat along dim 0 at the position of the
    first part, for kind in weight / bias (model_hf_llama.py:506-566)."""
    out = {}
    for k, t in state.items():
        head, _, kind = k.rpartition(".")
        if kind in ("weight", "bias") and head.endswith("." + parts[0]):
            prefix = head[: -len(parts[0])]
            names = [prefix + p + "." + kind for p in parts]
            for n in names[1:]:
                if n not in state:
                    raise KeyError(f"{k} has no partner {n}")
            if prefix + merged + "." + kind in state:
                raise KeyError(f"{prefix + merged}.{kind} already present")
            out[prefix + merged + "." + kind] = torch.cat([state[n] for n in names], dim=0)
        elif kind in ("weight", "bias") and any(head.endswith("." + p) for p in parts[1:]):
            continue
        else:
            out[k] = t
    return out


def preprocess_hf_llama(state: Mapping[str, torch.Tensor], n_heads: int, n_kv_heads: Optional[int], dim: int, rank: int = 0,
                        world: int = 1, mixtral: bool = False, router_row_parallel: bool = True) -> Dict[str, torch.Tensor]:
    """HF names -> this TP rank's tensors under the reference's final names (qkv_proj / gate_up_proj merged per rank).
    router_row_parallel: the reference shards Mixtral's router along its INPUT dimension and all-reduces the logits
    (model_hf_mixtral.py:157-160); chitu_amd.mixtral keeps the [experts, dim] matrix whole on every rank (False)."""
    n_kv = n_heads if n_kv_heads is None else n_kv_heads
    st = strip_model_prefix(state)
    if mixtral:
        st = map_mixtral_names(st)
    if world > 1:
        st = split_merged_gate_up(split_merged_qkv(st, n_heads, n_kv, dim // n_heads))
        row = HF_LLAMA_ROW + (("gate",) if mixtral and router_row_parallel else ())
        st = chunk_for_tensor_parallel(st, rank, world, column=HF_LLAMA_COLUMN, row=row)
    st = _merge_group(st, ("q_proj", "k_proj", "v_proj"), "qkv_proj")
    return _merge_group(st, ("gate_proj", "up_proj"), "gate_up_proj")


_LLAMA_MODULE_NAMES = (("embed_tokens.weight", "embed_weight"), ("lm_head.weight", "head_weight"), ("norm.weight", "norm"),
                       (".input_layernorm.weight", ".attn_norm"), (".post_attention_layernorm.weight", ".ffn_norm"),
                       (".self_attn.qkv_proj.weight", ".attn.wqkv"), (".self_attn.o_proj.weight", ".attn.wo"),
                       (".mlp.gate_up_proj.weight", ".ffn.w13"), (".mlp.down_proj.weight", ".ffn.w2"), (".mlp.gate.weight", ".ffn.gate"))


def _rename_llama(name: str) -> str:
    for a, b in _LLAMA_MODULE_NAMES:
        if name == a or (a.startswith(".") and name.endswith(a)):
            return name[: len(name) - len(a)] + b
    return name


def to_llama_module_names(state: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Reference names -> chitu_amd.llama.LlamaDecoder's parameters.  Biases (Qwen-style qkv bias) have no parameter there
    and are refused rather than dropped."""
    bias = [k for k in state if k.endswith(".bias")]
    if bias:
        raise NotImplementedError(f"LlamaDecoder has no bias parameters: {bias[:4]}")
    return {_rename_llama(k): v for k, v in state.items()}


def to_mixtral_module_names(state: Mapping[str, torch.Tensor], num_experts: int) -> Dict[str, torch.Tensor]:
    """Reference names -> chitu_amd.mixtral.MixtralDecoder's parameters: attention / norms / router as for Llama; the
    experts' bf16 gate_up / down matrices are quantised per output channel exactly as `simple_w8a8` does at load time
    (quantize/quantizer.py:117-145 -> quantize/w8a8.py:29-35: chitu_amd.quantize.w8a8.quant_weight, bit-exact against the
    reference's own function) and stacked [experts, ...] for the grouped int8 GEMMs."""
    from .quantize.w8a8 import quant_weight

    out, experts = {}, {}
    for k, v in state.items():
        m = re.match(r"(layers\.\d+)\.mlp\.experts\.(\d+)\.(gate_up_proj|down_proj)\.weight$", k)
        if m:
            experts.setdefault(m.group(1), {}).setdefault(m.group(3), {})[int(m.group(2))] = v
        else:
            out[_rename_llama(k)] = v
    for layer, mats in experts.items():
        for ref_name, mine in (("gate_up_proj", "w13"), ("down_proj", "w2")):
            per = mats.get(ref_name, {})
            if sorted(per) != list(range(num_experts)):
                raise KeyError(f"{layer}: experts present for {ref_name}: {sorted(per)}, expected 0..{num_experts - 1}")
            q = [quant_weight(per[e].to(torch.float16)) for e in range(num_experts)]  # the reference quantises its fp16 module weights
            out[f"{layer}.ffn.{mine}"] = torch.stack([w for w, _ in q])
            out[f"{layer}.ffn.{mine}_scale"] = torch.stack([s for _, s in q])
    return out


def load_checkpoint_hf_llama(model, path: str, rank: int = 0, world: int = 1, skip_preprocess: bool = False) -> None:
    """HF Llama-family directory -> a LlamaDecoder or MixtralDecoder built for this TP rank, one call.  skip_preprocess: the
    directory holds this rank's `model.rank{rank}.safetensors` written by `save_preprocessed` (script/preprocess_and_save.py's
    flow, backend.py:415-428) -- already sharded, merged, quantised and under module names."""
    from safetensors import safe_open

    args = model.args
    state = {}
    pattern = f"model.rank{rank}.safetensors" if skip_preprocess else "*.safetensors"
    files = sorted(glob.glob(os.path.join(path, pattern)))
    if not files:
        raise FileNotFoundError(f"no {pattern} under {path}")
    for fp in files:
        with safe_open(fp, framework="pt", device="cpu") as f:
            for name in f.keys():
                state[name] = f.get_tensor(name)
    if skip_preprocess:
        load_deepseek_v3(model, state)
        return
    mixtral = hasattr(args, "num_local_experts")
    st = preprocess_hf_llama(state, args.n_heads, args.n_kv_heads, args.dim, rank, world, mixtral=mixtral,
                             router_row_parallel=False)
    st = to_mixtral_module_names(st, args.num_local_experts) if mixtral else to_llama_module_names(st)
    load_deepseek_v3(model, st)  # the generic strict, shape- and dtype-checked in-place copy


# Meta-format Llama checkpoints (`consolidated.*.pth` names: tok_embeddings, attention.wq .. wo, feed_forward.w1 .. w3,
# attention_norm / ffn_norm, output) -- what the reference's TransformerLlama loads (models/model_llama.py:86-99), the class
# BASELINE config 1 runs.  Its column list spells the embedding "embed", which never matches "tok_embeddings"
# (utils.py:34-39), so at TP > 1 the reference leaves the table whole while its VocabParallelEmbedding expects a shard; here
# the table is sharded like every other vocabulary-parallel tensor.
META_LLAMA_COLUMN = ("wq", "wk", "wv", "w1", "w3", "output", "tok_embeddings")
META_LLAMA_ROW = ("wo", "w2")
_META_LLAMA_MODULE_NAMES = (("tok_embeddings.weight", "embed_weight"), ("output.weight", "head_weight"), ("norm.weight", "norm"),
                            (".attention_norm.weight", ".attn_norm"), (".ffn_norm.weight", ".ffn_norm"),
                            (".attention.wqkv.weight", ".attn.wqkv"), (".attention.wo.weight", ".attn.wo"),
                            (".feed_forward.w13.weight", ".ffn.w13"), (".feed_forward.w2.weight", ".ffn.w2"))


def preprocess_meta_llama(state: Mapping[str, torch.Tensor], rank: int = 0, world: int = 1) -> Dict[str, torch.Tensor]:
    """Meta names -> this TP rank's LlamaDecoder parameters: TP chunk (models/model.py:332-370), wq | wk | wv -> wqkv and
    w1 | w3 -> w13 per rank, module names."""
    st = chunk_for_tensor_parallel(state, rank, world, column=META_LLAMA_COLUMN, row=META_LLAMA_ROW)
    st = _merge_group(st, ("wq", "wk", "wv"), "wqkv")
    st = _merge_group(st, ("w1", "w3"), "w13")
    out = {}
    for k, v in st.items():
        for a, b in _META_LLAMA_MODULE_NAMES:
            if k == a or (a.startswith(".") and k.endswith(a)):
                k = k[: len(k) - len(a)] + b
                break
        out[k] = v
    return out
