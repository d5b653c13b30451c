"""ctypes binding of ``csrc/libfql_int4.so`` (C ABI: ``include/fql_int4.h``).

There is no fallback: if the shared library is missing or a symbol is absent the import of
this module's users fails loudly with instructions to build it.  ``torch`` must be imported
before the library is loaded so that the HIP runtime the library binds to is the one torch
already initialised (same SONAME ``libamdhip64.so.7``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import torch  # noqa: F401  (loads libamdhip64 first)

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.environ.get("FQL_INT4_LIB") or os.path.join(CSRC, "libfql_int4.so")   # env override: experiments only

PRECISION_DEFAULT = 0
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2
PRECISION_INT8 = 1
PRECISION_FAST = 2
PRECISION_EXACT = 3
PRECISION_FP8 = 8

_SYMBOLS = {
    # name: (restype, argtypes)
    "fql_version": (ctypes.c_int, []),
    "fql_error_string": (ctypes.c_char_p, [ctypes.c_int]),
    "fql_linear_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "fql_linear_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 4
                           + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_linear_bias_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4
                                + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_linear_group_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fql_moe_group_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int] * 5 + [ctypes.c_void_p]),
    "fql_group_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 6),
    "fql_linear_group_ws_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_moe_group_ws_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int] * 6 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_moe_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "fql_moe_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int] * 5
                        + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_moe_gather_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] + [ctypes.c_void_p] * 3
                               + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_moe_gather_scaled_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] + [ctypes.c_void_p] * 4
                                      + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_quantize_rows_f32": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_void_p]),
    "fql_quantize_tensor_f32": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_int] * 2 + [ctypes.c_void_p]),
    "fql_unpack_u8": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_dequantize_f32": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 2 + [ctypes.c_void_p]),
    "fql_act_padded_k": (ctypes.c_int, [ctypes.c_int]),
    "fql_act_limb_bytes": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "fql_act_quant_f32": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fql_gemm_i8_f32": (ctypes.c_int, [ctypes.c_void_p] * 9 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]),
    "fql_gemm_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "fql_moe_gated_fwd_f32": (ctypes.c_int, [ctypes.c_void_p] * 7 + [ctypes.c_int] * 5
                              + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_route_plan_i32": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 3 + [ctypes.c_void_p] * 5),
    "fql_combine_f32": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_void_p]),
    "fql_regroup_index_i32": (ctypes.c_int, [ctypes.c_void_p] + [ctypes.c_int] * 2 + [ctypes.c_void_p] * 5),
    "fql_native_dtype_supported": (ctypes.c_int, [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_int]),
    "fql_linear_fwd": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int] + [ctypes.c_void_p] * 4 + [ctypes.c_int] * 5
                       + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_moe_fwd_f8": (ctypes.c_int, [ctypes.c_void_p] * 8 + [ctypes.c_int] * 5 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_linear_fwd_f8": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "fql_moe_fwd": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 6
                    + [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
}

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC] + ([] if verbose else ["-s"])
    subprocess.check_call(cmd)
    return SO_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise NativeLibraryError(
            f"{SO_PATH} not found: the HIP extension is not built. "
            f"Run `make -C {CSRC}` (or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "There is no CPU fallback for GPU tensors.")
    try:
        handle = ctypes.CDLL(SO_PATH)
    except OSError as exc:  # pragma: no cover
        raise NativeLibraryError(f"cannot load {SO_PATH}: {exc}") from exc
    for name, (res, args) in _SYMBOLS.items():
        try:
            fn = getattr(handle, name)
        except AttributeError as exc:
            raise NativeLibraryError(f"{SO_PATH} does not export {name}; rebuild it") from exc
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def exported_symbols():
    return sorted(_SYMBOLS)


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().fql_error_string(rc).decode()
        raise RuntimeError(f"{what}: {msg} (code {rc})")
