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


def test_fused_launch_shape_limits_agree_between_python_and_c():
    """ops.bf16_add_norm_fits / ops.mla_q_proj_fits (what the models ask before taking a fused launch) and the C entry
    points' own limits are the same boundary: every shape the Python side calls unfit is refused by the library with
    CHITU_ERR_UNSUPPORTED before anything is launched (argument checks run on the host, so this needs no GPU; the
    pointers are never dereferenced)."""
    from chitu_amd import _lib, ops

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    buf = ctypes.create_string_buffer(64)
    p = ctypes.c_void_p(ctypes.addressof(buf))
    i32, i64, f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    unfit = 0
    for M in (1, 2, 3, 4, 5):
        for K in (256, 512, 520, 4096, 7168, 8192, 8256):
            for N in (16, 256, 6144, 100000):
                if ops.bf16_add_norm_fits(M, N, K):
                    continue
                unfit += 1
                rc = lib.chitu_hip_bf16_gemm_add_norm(p, i64(K // 8 * 8), p, i64(K // 8 * 8), i32(1), i64(0), p, i64(K // 8 * 8), p,
                                                      f32(1e-5), p, p, i32(0), i64(M), i64(N), i64(K), None)
                assert rc == -2, (M, N, K, rc)
                rc = lib.chitu_hip_bf16_gemm_silu_add_norm(p, i64(K // 8 * 8), p, i64(K // 8 * 8), p, i64(K // 8 * 8), p, f32(1e-5),
                                                           p, p, i64(M), i64(N), i64(K), None)
                assert rc == -2, (M, N, K, rc)
    assert unfit > 20
    for bs, ql in ((33, 1536), (64, 1536), (16, 2176), (16, 1600), (1, 4096)):
        assert not ops.mla_q_proj_fits(bs, ql)
        rc = lib.chitu_hip_mla_q_proj(p, i64(ql + 576 + (8 - (ql + 576) % 8) % 8), i32(ql), p, f32(1e-6), p, p, p, i32(0), i64(3072),
                                      p, f32(1e-6), p, p, p, i64(4), i32(64), p, i32(2), p, i32(bs), i32(512), i32(64), None)
        assert rc == -2, (bs, ql, rc)


def test_norm_prologue_launch_limits_agree_between_python_and_c():
    """Round 4's fused launch: ops.fp8_linear_add_norm_fits (attn_norm in the first projection's prologue) against
    chitu_hip_fp8_gemm_add_norm: an unfit shape is refused with CHITU_ERR_UNSUPPORTED on the host.  (A fit shape would
    launch: those are the GPU tests' business.)"""
    from chitu_amd import _lib, ops

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    buf = ctypes.create_string_buffer(64)
    p = ctypes.c_void_p(ctypes.addressof(buf))
    i32, i64, f32 = ctypes.c_int32, ctypes.c_int64, ctypes.c_float
    fit = unfit = 0
    for M in (1, 2, 3):
        for terms in (1, 2, 9, 16, 17):
            for N, K in ((2112, 7168), (3648, 2048), (4608, 7168), (832, 512), (16, 7168), (256, 16384), (2112, 7232 - 64), (64, 128),
                         (129280 // 8, 7168)):
                if K % 128 != 0:
                    continue
                if ops.fp8_linear_add_norm_fits(M, N, K, terms):
                    fit += 1
                    continue
                unfit += 1
                rc = lib.chitu_hip_fp8_gemm_add_norm(p, i64(K), p, i64(K * terms), i32(terms), i64(K), p, i64(K), p, f32(1e-6), p, p, p,
                                                     i32(0), i64(M), i64(N), i64(K), None)
                assert rc == -2, (M, terms, N, K, rc)
    assert fit >= 8 and unfit > 40
    assert ops.fp8_linear_add_norm_fits(1, 2112, 7168, 9) and ops.fp8_linear_add_norm_fits(1, 3648, 2048, 8)  # R1, V2-Lite at batch 1
    assert not ops.fp8_linear_add_norm_fits(2, 2112, 7168, 9) and ops.fp8_linear_add_norm_fits(2, 2112, 7168, 1)


def test_xcd_blocked_tile_order_covers_every_tile_once():
    """The prefill GEMM's workgroup -> tile map (gemm_common.h::xcd_tile_of), through its host-side entry: every tile of the
    grid exactly once, padding workgroups only beyond it, and the workgroups of one XCD (b % 8) inside ONE rectangle whose
    half-perimeter is the smallest an 8-way split allows."""
    import ctypes

    import numpy as np

    from chitu_amd import _lib

    lib = _lib.lib()
    for tiles_m, tiles_n in [(16, 17), (1, 1), (1, 56), (16, 24), (3, 5), (7, 129), (33, 2), (128, 4), (5, 8)]:
        grid = ctypes.c_int32()
        assert lib.chitu_hip_selftest_xcd_tile_order(tiles_m, tiles_n, ctypes.byref(grid), None, ctypes.c_int64(0)) == 0
        n = grid.value
        assert n >= tiles_m * tiles_n and n % 8 == 0
        tiles = np.full((n, 2), -7, dtype=np.int32)
        assert lib.chitu_hip_selftest_xcd_tile_order(tiles_m, tiles_n, ctypes.byref(grid),
                                                     tiles.ctypes.data_as(ctypes.c_void_p), ctypes.c_int64(n)) == 0
        real = tiles[tiles[:, 0] >= 0]
        assert len(real) == tiles_m * tiles_n and len({(a, b) for a, b in real.tolist()}) == tiles_m * tiles_n
        assert real[:, 0].max() == tiles_m - 1 and real[:, 1].max() == tiles_n - 1
        assert (tiles[tiles[:, 0] < 0] == -1).all()
        best = min((tiles_m + xm - 1) // xm + (tiles_n + 8 // xm - 1) // (8 // xm) for xm in (1, 2, 4, 8))
        for xcd in range(8):
            mine = tiles[xcd::8]
            mine = mine[mine[:, 0] >= 0]
            if len(mine):
                span = (mine[:, 0].max() - mine[:, 0].min() + 1) + (mine[:, 1].max() - mine[:, 1].min() + 1)
                assert span <= best, (tiles_m, tiles_n, xcd, span, best)
