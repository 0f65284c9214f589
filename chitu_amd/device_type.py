"""Vendor seam.  Mirrors chitu/device_type.py:13-20 and adds `is_amd()`.

The reference dispatches on substrings of the device name (fused_moe.py:605,
model_deepseek_v3.py:85,935,968).  On MI355X both is_nvidia() and is_muxi() are False.
"""

import torch

_device_name = None


def get_device_name():
    global _device_name
    if _device_name is None:
        _device_name = torch.cuda.get_device_name() if torch.cuda.is_available() else "cpu"
    return _device_name


def is_nvidia():
    return "NVIDIA" in get_device_name()


def is_muxi():
    name = get_device_name()
    return any(p in name for p in ("4000", "4001"))


def is_amd():
    name = get_device_name()
    return ("AMD" in name) or ("MI3" in name) or ("Instinct" in name)
