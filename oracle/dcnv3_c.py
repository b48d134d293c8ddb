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


class _Geom(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("kh", "kw", "sh", "sw", "ph", "pw", "dh", "dw", "G", "D", "rc")]


def _geom(kh, kw, sh, sw, ph, pw, dh, dw, G, D, rc):
    return _Geom(kh, kw, sh, sw, ph, pw, dh, dw, G, D, rc)


def _out_hw(H, W, g):
    return ((H + 2 * g.ph - (g.dh * (g.kh - 1) + 1)) // g.sh + 1, (W + 2 * g.pw - (g.dw * (g.kw - 1) + 1)) // g.sw + 1)


def dcnv3_forward_any_c(inp, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, G, D, offset_scale, remove_center=0):
    """float64, independent h / w geometry, any D (dcnv3_ref.c: dcnv3_forward_any_c)."""
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "dcnv3_ref.c")):
        build()
    lib = ctypes.CDLL(_SO)
    inp = np.ascontiguousarray(inp, dtype=np.float64)
    offset = np.ascontiguousarray(offset, dtype=np.float64).reshape(-1)
    mask = np.ascontiguousarray(mask, dtype=np.float64).reshape(-1)
    N, H, W, C = inp.shape
    g = _geom(kh, kw, sh, sw, ph, pw, dh, dw, G, D, remove_center)
    Ho, Wo = _out_hw(H, W, g)
    out = np.empty((N, Ho, Wo, C), dtype=np.float64)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.dcnv3_forward_any_c(inp.ctypes.data_as(dp), offset.ctypes.data_as(dp), mask.ctypes.data_as(dp), out.ctypes.data_as(dp),
                                 N, H, W, ctypes.byref(g), ctypes.c_double(offset_scale))
    assert rc == 0
    return out


def dcnv3_backward_any_c(inp, offset, mask, grad_out, kh, kw, sh, sw, ph, pw, dh, dw, G, D, offset_scale, remove_center=0):
    """float64 backward: returns (grad_input, grad_offset, grad_mask) shaped like input / offset / mask."""
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "dcnv3_ref.c")):
        build()
    lib = ctypes.CDLL(_SO)
    inp = np.ascontiguousarray(inp, dtype=np.float64)
    off = np.ascontiguousarray(offset, dtype=np.float64)
    msk = np.ascontiguousarray(mask, dtype=np.float64)
    go = np.ascontiguousarray(grad_out, dtype=np.float64)
    N, H, W, C = inp.shape
    g = _geom(kh, kw, sh, sw, ph, pw, dh, dw, G, D, remove_center)
    gi, goff, gm = np.zeros_like(inp), np.zeros_like(off), np.zeros_like(msk)
    dp = ctypes.POINTER(ctypes.c_double)
    rc = lib.dcnv3_backward_any_c(inp.ctypes.data_as(dp), off.ctypes.data_as(dp), msk.ctypes.data_as(dp), go.ctypes.data_as(dp),
                                  gi.ctypes.data_as(dp), goff.ctypes.data_as(dp), gm.ctypes.data_as(dp), N, H, W, ctypes.byref(g),
                                  ctypes.c_double(offset_scale))
    assert rc == 0
    return gi, goff, gm
