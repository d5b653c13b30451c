"""ctypes loader for the plain-C oracle (oracle/libint4_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libint4_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _p(a, t):
    return a.ctypes.data_as(t)


def quantize_rows(w):
    w = np.ascontiguousarray(w, dtype=np.float32)
    N, K = w.shape
    packed = np.empty((N, K // 2), np.uint8)
    s = np.empty(N, np.float32)
    z = np.empty(N, np.float32)
    rc = lib().oracle_quantize_rows(_p(w, _f32p), N, K, _p(packed, _u8p), _p(s, _f32p), _p(z, _f32p))
    assert rc == 0
    return packed, s, z


def quantize_tensor(w):
    w = np.ascontiguousarray(w, dtype=np.float32)
    N, K = w.shape
    packed = np.empty((N, K // 2), np.uint8)
    s = ctypes.c_float()
    z = ctypes.c_float()
    rc = lib().oracle_quantize_tensor(_p(w, _f32p), N, K, _p(packed, _u8p), ctypes.byref(s), ctypes.byref(z))
    assert rc == 0
    return packed, np.float32(s.value), np.float32(z.value)


def unpack(packed):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    q = np.empty(packed.shape[:-1] + (packed.shape[-1] * 2,), np.uint8)
    lib().oracle_unpack(_p(packed, _u8p), ctypes.c_size_t(packed.size), _p(q, _u8p))
    return q


def dequantize(packed, scales, zps):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    N, K2 = packed.shape
    w = np.empty((N, K2 * 2), np.float32)
    lib().oracle_dequantize(_p(packed, _u8p), _p(np.ascontiguousarray(scales, np.float32), _f32p),
                            _p(np.ascontiguousarray(zps, np.float32), _f32p), N, K2 * 2, _p(w, _f32p))
    return w


def _linear(fn, out_dtype, outp, x, packed, scales, zps):
    x = np.ascontiguousarray(x, dtype=np.float32)
    squeeze = x.ndim == 1
    if squeeze:
        x = x[None]
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    N, K2 = packed.shape
    B = x.shape[0]
    out = np.empty((B, N), out_dtype)
    fn(_p(x, _f32p), _p(packed, _u8p), _p(np.ascontiguousarray(scales, np.float32), _f32p),
       _p(np.ascontiguousarray(zps, np.float32), _f32p), B, K2 * 2, N, _p(out, outp))
    return out[0] if squeeze else out


def linear_f64acc(x, packed, scales, zps):
    return _linear(lib().oracle_linear_f64acc, np.float64, _f64p, x, packed, scales, zps)


def linear_fma(x, packed, scales, zps):
    return _linear(lib().oracle_linear_fma, np.float32, _f32p, x, packed, scales, zps)


def moe_grouped(packed, scales, zps, inputs, tokens_per_expert, input_offsets):
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    E, N, K2 = packed.shape
    inputs = np.ascontiguousarray(inputs, dtype=np.float32)
    T = inputs.shape[0]
    out = np.empty((T, N), np.float64)
    rc = lib().oracle_moe_grouped(
        _p(packed, _u8p), _p(np.ascontiguousarray(scales, np.float32), _f32p),
        _p(np.ascontiguousarray(zps, np.float32), _f32p), _p(inputs, _f32p),
        _p(np.ascontiguousarray(tokens_per_expert, np.int32), _i32p),
        _p(np.ascontiguousarray(input_offsets, np.int32), _i32p), E, T, K2 * 2, N, _p(out, _f64p))
    assert rc == 0, rc
    return out
