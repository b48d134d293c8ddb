"""ORACLE (test infrastructure): ctypes binding of oracle/dcnv3_ref.c."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdcnv3_ref.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def dcnv3_forward_c(inp, offset, mask, K, stride, pad, dil, G, D, offset_scale, remove_center=0):
    if not os.path.exists(_SO):
        build()
    lib = ctypes.CDLL(_SO)
    inp = np.ascontiguousarray(inp, dtype=np.float32)
    offset = np.ascontiguousarray(offset, dtype=np.float32).reshape(-1)
    mask = np.ascontiguousarray(mask, dtype=np.float32).reshape(-1)
    N, H, W, C = inp.shape
    Ho = (H + 2 * pad - (dil * (K - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (K - 1) + 1)) // stride + 1
    out = np.empty((N, Ho, Wo, C), dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    rc = lib.dcnv3_forward_c(inp.ctypes.data_as(fp), offset.ctypes.data_as(fp), mask.ctypes.data_as(fp),
                             out.ctypes.data_as(fp), N, H, W, G, D, K, stride, pad, dil,
                             ctypes.c_float(offset_scale), remove_center)
    assert rc == 0
    return out
