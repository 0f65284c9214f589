"""Checkpoint preprocessing for the DeepSeek-V3 / R1 decode path: Hugging Face names -> this
package's module tree, tensor-parallel shards, merged / stacked FP8 tensors, per-rank preprocessed files.

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
