"""Import the reference's Python modules (read-only, /root/reference) on CPU.

Only used by gen_golden.py in the build container -- /root/reference does not exist
on the GPU box, so nothing in tests/ imports this at test time.  Recipe follows
SURVEY.md section 8(c): Triton interpreter, stub `chitu_backend`/`tiktoken`, no-op
cuda sync, device name "cpu".
"""

import os
import sys
import types

REFERENCE_ROOT = "/root/reference"


def install():
    os.environ.setdefault("TRITON_INTERPRET", "1")
    os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
    sys.dont_write_bytecode = True
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    for name in ("chitu_backend", "tiktoken", "tiktoken.load"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "tiktoken.load":
                m.load_tiktoken_bpe = lambda *a, **k: {}
            sys.modules[name] = m
    import torch

    torch.cuda.synchronize = lambda *a, **k: None
    import chitu.device_type as dt

    dt._device_name = "cpu"
    return dt
