"""Source-compatible stand-in for the reference's pybind module `chitu_backend`
(csrc/binding.cpp:16-19): exports `cuda_moe_align_block_size` with the same 7 arguments
(csrc/moe_kernel.h:7-11), implemented by chitu_hip_moe_align_block_size.

    import chitu_amd.chitu_backend as chitu_backend      # in chitu/fused_moe.py:20
"""

from .fused_moe import cuda_moe_align_block_size  # noqa: F401

__all__ = ["cuda_moe_align_block_size"]
__doc__ = (__doc__ or "") + "\nA Supa Fast inference engine (MI355X path)."
