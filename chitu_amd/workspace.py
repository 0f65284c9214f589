"""Persistent device scratch for split-K partials and MoE intermediates.

Allocated once per device and reused, so graph-captured launches always see the same
addresses (the reference allocates scratch per call, e.g. fused_moe.py:1176-1187,
attn_backend.py:737-746 -- that is not capture-safe without the caching allocator's pool).
"""

import torch

_ws = {}
_retired = []  # outgrown buffers stay alive: hipGraphs captured earlier still launch on their addresses
_namespace = None


def set_namespace(ns):
    """Scratch is per (device, tag) -- one decode engine per process and device, the normal deployment.  Several engines
    driven from ONE process on one device at the same time (the tests' rank-instances on separate streams) each select
    their own namespace before they launch or capture, so that their launches never share scratch."""
    global _namespace
    _namespace = ns


def get(nbytes: int, device, tag: str = "default") -> torch.Tensor:
    key = (torch.device(device).index or 0, tag, _namespace)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            # a buffer created under capture would live in that graph's PRIVATE pool (and die with it, while this
            # table and later graphs still hold its address); growth would change an address earlier launches of the
            # same capture already recorded.  decode() runs one eager step before it captures, so neither happens on
            # the product path -- refuse both instead of relying on that.
            raise RuntimeError(
                f"workspace '{tag}' must be {'created' if buf is None else f'grown ({buf.numel()} -> {nbytes} B)'} "
                "during graph capture; run one eager step first"
            )
        if buf is not None:
            _retired.append(buf)
        size = max(nbytes, 1 << 20)
        buf = torch.empty(size, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf
