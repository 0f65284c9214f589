"""chitu_amd -- MI355X (gfx950) native decode hot path behind thu-pacman/chitu's operator surface.

Modules mirror the reference's names so `chitu.ops` / `chitu.fused_moe` / `chitu.attn_backend` /
`chitu.cache_manager` / `chitu.tensor_parallel` call sites can import from here unchanged
(see INTEGRATION.md).  All device work goes through the C-ABI library libchitu_hip.so
(include/chitu_hip.h); there is no CPU or eager-PyTorch fallback for any op.
"""

__version__ = "0.1.0"
