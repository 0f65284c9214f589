#!/usr/bin/env python3
"""Dump the call signatures of the reference's operator surface -> tests/golden/ref_signatures.json.

Run in the build container (needs /root/reference):  python tests/golden/gen_signatures.py
For every public function, class and public method of the reference modules that chitu_amd mirrors, the ordered
parameter list with kinds and defaults (repr).  tests/test_reference_signatures.py holds chitu_amd to it.
"""
import inspect
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_shims  # noqa: E402

ref_shims.install()
import types  # noqa: E402

for closed in ("w8a8gemm", "w8a8gemv"):  # closed third-party kernels of chitu/quantize/w8a8.py: only the signatures matter here
    sys.modules.setdefault(closed, types.ModuleType(closed))

MODULES = {  # reference module -> chitu_amd module
    "chitu.ops": "chitu_amd.ops",
    "chitu.fused_moe": "chitu_amd.fused_moe",
    "chitu.tensor_parallel": "chitu_amd.tensor_parallel",
    "chitu.cache_manager": "chitu_amd.cache_manager",
    "chitu.attn_backend": "chitu_amd.attn_backend",
    "chitu.quantize.w8a8": "chitu_amd.quantize.w8a8",
    "chitu.device_type": "chitu_amd.device_type",
}


def unwrap(fn):
    """The reference wraps its Triton ops in `auto_retry_triton_compilation` (chitu/ops.py:10-43), a plain closure
    without functools.wraps: the op itself is the function held in the wrapper's closure."""
    while inspect.isfunction(fn) and fn.__name__ == "wrapped" and fn.__closure__:
        inner = [c.cell_contents for c in fn.__closure__ if inspect.isfunction(c.cell_contents)]
        if not inner:
            break
        fn = inner[0]
    return fn


def sig(fn):
    fn = unwrap(fn)
    try:
        s = inspect.signature(fn)
    except (TypeError, ValueError):
        return None
    return [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)] for p in s.parameters.values()]


def main():
    import importlib

    out = {}
    for ref_name, ours in MODULES.items():
        try:
            mod = importlib.import_module(ref_name)
        except Exception as e:  # noqa: BLE001
            print(f"skip {ref_name}: {e!r}")
            continue
        entry = {}
        for name, obj in vars(mod).items():
            if name.startswith("_") or getattr(unwrap(obj), "__module__", None) != ref_name:
                continue
            if inspect.isfunction(obj):
                s = sig(obj)
                if s is not None:
                    entry[name] = s
            elif inspect.isclass(obj):
                for mname, m in vars(obj).items():
                    if (mname == "__init__" or not mname.startswith("_")) and (inspect.isfunction(m) or isinstance(m, (staticmethod, classmethod))):
                        f = m.__func__ if isinstance(m, (staticmethod, classmethod)) else m
                        s = sig(f)
                        if s is not None:
                            entry[f"{name}.{mname}"] = s
        out[ref_name] = {"mirror": ours, "symbols": entry}
        print(ref_name, len(entry), "symbols")
    json.dump(out, open(os.path.join(HERE, "ref_signatures.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
