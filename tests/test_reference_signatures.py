"""The drop-in boundary by signature: every symbol of the reference's operator surface that chitu_amd
mirrors takes the reference's parameters -- same names, order, kinds and defaults.  chitu_amd may EXTEND a
signature with trailing defaulted parameters (a call written against the reference never sees them); it may not
rename, reorder or drop one.

The reference's side is a fixture (tests/golden/ref_signatures.json) written by tests/golden/gen_signatures.py
from the imported reference; where /root/reference is present (the build container) the fixture is regenerated
in a subprocess and must not have drifted."""

import importlib
import inspect
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "ref_signatures.json")

# reference symbol -> chitu_amd symbol where the NAME differs: one HIP backend stands for the reference's
# per-vendor attention backends (flash_attn for GQA, Triton / FlashMLA / FlashInfer for MLA)
ALIASES = {
    ("chitu.attn_backend", "FlashAttnBackend.attn_varlen_func"): "HipAttnBackend.attn_varlen_func",
    ("chitu.attn_backend", "FlashAttnBackend.attn_with_kvcache"): "HipAttnBackend.attn_with_kvcache",
    ("chitu.attn_backend", "TritonAttnBackend.prepare_metadata_for_decode"): "HipAttnBackend.prepare_metadata_for_decode",
    ("chitu.attn_backend", "TritonAttnBackend.mla_attn_with_kvcache"): "HipAttnBackend.mla_attn_with_kvcache",
}

# the symbols the decode path is called through: these MUST exist on our side
REQUIRED = {
    "chitu.ops": ["append_to_paged_kv_cache", "apply_rotary_pos_emb", "act_quant_deepseek_v3", "weight_dequant_deepseek_v3",
                  "weight_dequant_soft_fp8_deepseek_v3", "fp8_gemm_deepseek_v3", "soft_fp8_gemm_deepseek_v3"],
    "chitu.fused_moe": ["moe_align_block_size", "fused_experts", "fused_experts_impl", "per_token_group_quant_fp8",
                        "SiluAndMul.forward"],
    "chitu.tensor_parallel": ["init_tp", "get_tp_group", "get_tp_size", "ColumnParallelLinear.__init__",
                              "ColumnParallelLinear.forward", "RowParallelLinear.__init__", "RowParallelLinear.forward",
                              "VocabParallelEmbedding.__init__", "VocabParallelEmbedding.forward"],
    "chitu.cache_manager": [f"PagedKVCacheManager.{m}" for m in (
        "__init__", "get_gpu_block_table", "get_gpu_seq_lens_excl_this_decode", "get_gpu_seq_lens_incl_this_decode",
        "get_block_size", "get_paged_kv_cache", "prepare_cache_decode", "prepare_block_table_for_decode",
        "finalize_cache_bylayer_prefill", "finalize_cache_all_prefill", "finalize_cache_single_decode",
        "finalize_cache_all_decode", "free_req_cache_blocks")],
    "chitu.attn_backend": ["AttnBackend.attn_varlen_func", "AttnBackend.attn_with_kvcache", "AttnBackend.prepare_metadata_for_decode",
                           "FlashAttnBackend.attn_varlen_func", "FlashAttnBackend.attn_with_kvcache",
                           "TritonAttnBackend.prepare_metadata_for_decode", "TritonAttnBackend.mla_attn_with_kvcache"],
    "chitu.quantize.w8a8": ["quant_act", "quant_weight", "W8A8Linear.__init__", "W8A8Linear.forward"],
    "chitu.device_type": ["is_nvidia", "is_muxi"],
}


def _sig(fn):
    s = inspect.signature(fn)
    return [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)] for p in s.parameters.values()]


def _resolve(mod, dotted):
    obj = mod
    for part in dotted.split("."):
        if not hasattr(obj, part):
            return None
        obj = getattr(obj, part)
    return obj


def _fixture():
    return json.load(open(FIXTURE))


def test_mirrored_symbols_take_the_reference_parameters():
    fx = _fixture()
    checked, problems = 0, []
    for ref_mod, entry in fx.items():
        ours = importlib.import_module(entry["mirror"])
        for name, want in entry["symbols"].items():
            obj = _resolve(ours, ALIASES.get((ref_mod, name), name))
            if obj is None:
                if name in REQUIRED.get(ref_mod, []):
                    problems.append(f"{ref_mod}.{name}: missing in {entry['mirror']}")
                continue
            got = _sig(obj)
            checked += 1
            if got[: len(want)] != want:
                problems.append(f"{ref_mod}.{name}: reference {want} vs ours {got}")
            elif any(p[2] is None and p[1] not in ("VAR_POSITIONAL", "VAR_KEYWORD") for p in got[len(want):]):
                problems.append(f"{ref_mod}.{name}: extra parameters without defaults {got[len(want):]}")
    assert not problems, "\n".join(problems)
    assert checked >= 45, checked


def test_required_symbols_are_in_the_fixture():
    fx = _fixture()
    for mod, names in REQUIRED.items():
        for n in names:
            assert n in fx[mod]["symbols"], (mod, n)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference lives in the build container only")
def test_fixture_matches_the_reference_tree(tmp_path):
    before = _fixture()
    res = subprocess.run([sys.executable, os.path.join(HERE, "golden", "gen_signatures.py")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    assert _fixture() == before, "tests/golden/ref_signatures.json drifted from /root/reference: regenerate and review"
