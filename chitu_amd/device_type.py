"""Vendor seam of the operator surface: which device family are we on?

The reference dispatches on substrings of the CUDA device name (chitu/device_type.py:13-20, consulted at
fused_moe.py:605 and model_deepseek_v3.py:85,935,968).  The same three predicates exist here under the same
names, plus `is_amd()`; on an MI355X `is_nvidia()` and `is_muxi()` are both False, so reference call sites that
import this module fall through to the HIP path.  One table, one cached lookup.
"""

import functools

import torch

# family -> substrings of torch.cuda.get_device_name() that identify it
_FAMILIES = {
    "nvidia": ("NVIDIA",),
    "muxi": ("4000", "4001"),
    "amd": ("AMD", "MI3", "Instinct"),
}


@functools.lru_cache(maxsize=None)
def get_device_name() -> str:
    """Name of the current accelerator ("cpu" where there is none: host-side tests import this module)."""
    return torch.cuda.get_device_name() if torch.cuda.is_available() else "cpu"


def _is(family: str) -> bool:
    name = get_device_name()
    return any(tag in name for tag in _FAMILIES[family])


def is_nvidia() -> bool:
    return _is("nvidia")


def is_muxi() -> bool:
    return _is("muxi")


def is_amd() -> bool:
    return _is("amd")
