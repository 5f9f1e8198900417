"""ctypes wrapper of oracle/_build/libqt_oracle.so (C restatement of the streaming pass).
TEST INFRASTRUCTURE ONLY -- see qt_oracle.c / qt_oracle.py headers."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libqt_oracle.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.run(["make", "-C", _HERE], check=True)
        L = ctypes.CDLL(_SO)
        L.qto_num_threads.restype = ctypes.c_int
        L.qto_amax_bf16.restype = ctypes.c_uint16
        L.qto_amax_bf16.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.qto_amax_f32.restype = ctypes.c_float
        L.qto_amax_f32.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        L.qto_fake_quant_bf16.restype = None
        L.qto_fake_quant_bf16.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_uint16]
        L.qto_fake_quant_f32.restype = None
        L.qto_fake_quant_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_float]
        _lib = L
    return _lib


def num_threads():
    return lib().qto_num_threads()


def fake_quant_bf16(xbits, qmap, scale_bits, out=None):
    x = np.ascontiguousarray(xbits, dtype=np.uint16)
    q = np.ascontiguousarray(qmap, dtype=np.uint16)
    y = np.empty_like(x) if out is None else out
    lib().qto_fake_quant_bf16(x.ctypes.data, y.ctypes.data, x.size, q.ctypes.data, int(scale_bits))
    return y


def fake_quant_f32(x, qmap, scale, out=None):
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.ascontiguousarray(qmap, dtype=np.uint16)
    y = np.empty_like(x) if out is None else out
    lib().qto_fake_quant_f32(x.ctypes.data, y.ctypes.data, x.size, q.ctypes.data, float(scale))
    return y


def amax_bf16(xbits):
    x = np.ascontiguousarray(xbits, dtype=np.uint16)
    return int(lib().qto_amax_bf16(x.ctypes.data, x.size))


def amax_f32(x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    return float(lib().qto_amax_f32(x.ctypes.data, x.size))
