"""ctypes binding of libchitu_hip.so (the C-ABI in include/chitu_hip.h).

There is deliberately no fallback: if the HIP library is missing or a call fails,
the op raises.  PyTorch is used only for device memory and the current stream.
"""

import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHITU_HIP_LIB: load another build of the same C-ABI instead (A/B timing of two builds on one GPU)
LIB_PATH = os.environ.get("CHITU_HIP_LIB") or os.path.join(_HERE, "libchitu_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


class HipCallError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile every .hip under csrc/ for gfx950 and link libchitu_hip.so (in-tree)."""
    cmd = ["make", "-C", CSRC_DIR, "-j", str(os.cpu_count() or 4)]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
        print(res.stderr)
    if res.returncode != 0:
        raise RuntimeError("building libchitu_hip.so failed")
    return LIB_PATH


# Diagnostics (tests, chitu_amd.graphs.capture_verified in its diagnostic mode): while this is a list, every C-ABI call
# is appended to it as (entry name, (argument values ...)) -- device pointers, sizes, the stream -- before it is made.
call_log = None
# The same with the ctypes argument objects kept as they were passed, so that a recorded launch can be issued again (bench.py's
# per-kernel probes replay one step's launches of a kernel back to back in a hipGraph; the stream argument -- always the last --
# is replaced by the replaying stream).
call_log_ctypes = None


class _Recorder:
    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)

        def recorded(*args):
            if call_log is not None:
                call_log.append((name, tuple(getattr(a, "value", a) for a in args)))
            if call_log_ctypes is not None:
                call_log_ctypes.append((name, args))
            return fn(*args)

        return recorded


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipLibraryMissing(
                f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C chitu_amd/csrc`). There is no CPU fallback."
            )
        _lib = ctypes.CDLL(LIB_PATH)
    return _lib if call_log is None and call_log_ctypes is None else _Recorder(_lib)


def ptr(t):
    """Device (or host) address of a tensor's first element; None -> NULL."""
    if t is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    """Raw hipStream_t of torch's current stream, so launches are graph-capturable."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(status: int, what: str):
    if status != 0:
        if status > 0:
            raise HipCallError(f"{what}: HIP error {status}")
        raise HipCallError(f"{what}: bad argument (code {status})")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipCallError(
                "chitu_amd ops run only on a ROCm device tensor (no CPU fallback); got a CPU tensor"
            )


_INT_DTYPE_CODE = {
    torch.uint8: 0,
    torch.int8: 1,
    torch.int16: 2,
    torch.int32: 3,
    torch.int64: 4,
}

# activation / output element codes shared by the C-ABI
DT_BF16 = 0
DT_F16 = 1
DT_F32 = 2
_FLOAT_DTYPE_CODE = {torch.bfloat16: DT_BF16, torch.float16: DT_F16, torch.float32: DT_F32}


def int_dtype_code(dt):
    try:
        return _INT_DTYPE_CODE[dt]
    except KeyError:
        raise HipCallError(f"unsupported integer dtype {dt}")


def float_dtype_code(dt):
    try:
        return _FLOAT_DTYPE_CODE[dt]
    except KeyError:
        raise HipCallError(f"unsupported float dtype {dt}")


def i64(v):
    return ctypes.c_int64(int(v))


def i32(v):
    return ctypes.c_int32(int(v))


def f32(v):
    return ctypes.c_float(float(v))


DEBUG_OPTIONS = {"moe_gemm1_wk": 0, "moe_gemm1_nw": 1, "moe_gemm1_d": 2, "moe_gemm2_cfg": 3, "moe_i8_wk": 4,
                 "gate_generic": 5, "gate_ticket": 6, "sample_radix": 7, "fp8_gemm_wk": 8, "fp8_gemm_deep": 9,
                 "bf16_gemm_wk": 10, "bf16_gemm_deep": 11, "bf16_silu_wk": 12, "fp8_gemm_tiled": 13, "bf16_gemm_tiled": 14,
                 "gate_small_sort": 15, "fp8_tiled_tm": 16}


def apply_debug_options_from_env():
    """Tools only (tools/run_extra.py, tools/bench_kernels.py): CHITU_DEBUG_OPTIONS="name=value,..." forces launch
    variants for a sweep.  The product never calls this."""
    for kv in filter(None, os.environ.get("CHITU_DEBUG_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        check(lib().chitu_hip_debug_option(i32(DEBUG_OPTIONS[k]), i32(int(v))), "debug_option")


class debug_option:
    """`with debug_option("gate_generic", 1): ...` forces a launch variant of identical results
    (chitu_hip_debug_option) for equivalence tests and tuning sweeps; restored to the heuristic on exit."""

    def __init__(self, name: str, value: int):
        self.opt, self.value = DEBUG_OPTIONS[name], int(value)

    def __enter__(self):
        check(lib().chitu_hip_debug_option(i32(self.opt), i32(self.value)), "debug_option")
        return self

    def __exit__(self, *exc):
        check(lib().chitu_hip_debug_option(i32(self.opt), i32(-1)), "debug_option")
        return False
