"""The C-ABI library loads (no GPU needed) and exports every symbol include/chitu_hip.h declares."""

import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "chitu_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(chitu_hip_\w+)\s*\(", src)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "chitu_hip_moe_align_block_size" in syms
    assert len(syms) >= 7


def test_library_exports_every_declared_symbol():
    from chitu_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


def test_every_exported_entry_point_is_declared():
    """No undocumented entry points: `nm -D` of the library vs the header."""
    import subprocess

    from chitu_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = sorted(set(re.findall(r"\bT\s+(chitu_hip_\w+)", out)))
    assert exported == declared_symbols()


def test_ops_fail_loudly_without_device_tensors():
    """Product path has no CPU fallback: CPU tensors are refused, not silently computed."""
    import torch

    from chitu_amd import fused_moe, ops
    from chitu_amd._lib import HipCallError

    with pytest.raises(HipCallError):
        fused_moe.moe_align_block_size(torch.zeros(4, 2, dtype=torch.int64), 4, 4)
    with pytest.raises(HipCallError):
        ops.act_quant_deepseek_v3(torch.zeros(2, 128, dtype=torch.bfloat16))
