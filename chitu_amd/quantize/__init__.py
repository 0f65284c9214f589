"""Quantised linears on the decode path (reference: chitu/quantize/).  Only the W8A8 int8 path
(`simple_w8a8`, quantizer.py:117-145) is on the MI355X hot path; AWQ / GPTQ / EETQ / muxi variants are
other formats backed by closed or third-party kernels and are out of scope."""

from .w8a8 import W8A8Linear, quant_act, quant_weight, replace_with_simple_w8a8  # noqa: F401


def quant(model, method="simple_w8a8", **kwargs):
    """Dispatcher with the reference's name (chitu/quantize/quantizer.py:277-291)."""
    if method in ("simple_w8a8", "w8a8"):
        return replace_with_simple_w8a8(model, **kwargs)
    raise NotImplementedError(f"quant method {method!r} is not part of the MI355X decode path")
