"""Shared test helpers (fixture loading, bit views)."""

import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def bf16(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).view(torch.bfloat16)


def fp8(bits):
    return torch.from_numpy(np.ascontiguousarray(bits)).view(torch.float8_e4m3fn)


def bits16(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


def bits8(t):
    return t.detach().cpu().contiguous().view(torch.uint8).numpy()


def rel_err(a, b):
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-12)).item()


def max_rel_to_peak(a, b):
    """max |a-b| / max|b|: the '<=1e-2 rel' bar of BASELINE.md for fp8 GEMM / attention."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def pattern_cache(pages, page, dim):
    p = torch.arange(pages).view(-1, 1, 1)
    s = torch.arange(page).view(1, -1, 1)
    d = torch.arange(dim).view(1, 1, -1)
    return (((p * 64 + s) % 251).float() * 0.25 + (d % 7).float() - 3.0).to(torch.bfloat16)
