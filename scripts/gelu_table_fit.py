"""Coefficients of the 64-segment piecewise-quadratic Phi(x) table used by gelu_tab (csrc/common.hpp): Phi(x) ~ c0 + fr (c1 + fr c2) on
segment i = floor(u), fr = u - i, u = (x + 4.4) * 64 / 8.8, x clamped to [-4.4, 4.4).  c1, c2 are stored as fp16 (one packed dword per
segment, fetched with ONE ds_bpermute), c0 as fp32; after rounding c1 and c2 the fit is redone for the remaining free coefficients, so
the fp16 rounding costs nothing measurable.  Prints the C arrays and the max |GELU error| (float64 reference, fp32 evaluation order)."""
import numpy as np
from math import erf, sqrt
NSEG, LO, HI = 64, -4.4, 4.4
h = (HI - LO) / NSEG
phi = np.vectorize(lambda x: 0.5 * (1.0 + erf(x / sqrt(2.0))))
c0s, c1s, c2s = [], [], []
for i in range(NSEG):
    fr = np.linspace(0.0, 1.0, 4001)
    x = LO + (i + fr) * h
    y = phi(x)
    wgt = np.maximum(np.abs(x), 0.25)                       # the GELU error is |x| times the Phi error
    A = np.stack([np.ones_like(fr), fr, fr * fr], 1) * wgt[:, None]
    c = np.linalg.lstsq(A, y * wgt, rcond=None)[0]
    c2 = np.float16(c[2]).astype(np.float64)
    A1 = np.stack([np.ones_like(fr), fr], 1) * wgt[:, None]
    c01 = np.linalg.lstsq(A1, (y - c2 * fr * fr) * wgt, rcond=None)[0]
    c1 = np.float16(c01[1]).astype(np.float64)
    c0 = np.average(y - c1 * fr - c2 * fr * fr, weights=wgt * wgt)
    # minimax polish of c0 (centre the error band)
    r = (y - (c0 + c1 * fr + c2 * fr * fr)) * np.abs(x)
    c0 += 0.5 * (r.max() + r.min()) / max(np.abs(x).mean(), 1e-3) if np.abs(x).mean() > 0.3 else 0.5 * ((y - (c0 + c1 * fr + c2 * fr * fr)).max() + (y - (c0 + c1 * fr + c2 * fr * fr)).min())
    c0s.append(np.float32(c0)); c1s.append(np.float16(c1)); c2s.append(np.float16(c2))
c0s, c1s, c2s = np.array(c0s, np.float32), np.array(c1s, np.float16), np.array(c2s, np.float16)
# evaluate in fp32, the kernel's operation order
xs = np.linspace(-6.0, 6.0, 2000001).astype(np.float32)
xc = np.clip(xs, np.float32(LO), np.nextafter(np.float32(HI), np.float32(0)))
S = np.float32(NSEG / (HI - LO))
u = xc * S + np.float32(NSEG / 2)
idx = np.minimum(u.astype(np.int32), NSEG - 1)
fr = (u - np.floor(u)).astype(np.float32)
q = (c2s[idx].astype(np.float32) * fr + c1s[idx].astype(np.float32)).astype(np.float32)
P = (q * fr + c0s[idx]).astype(np.float32)
g = (xs * P).astype(np.float32)
ref = xs.astype(np.float64) * phi(xs.astype(np.float64))
err = np.abs(g.astype(np.float64) - ref)
print(f"// max |gelu_tab - GELU| = {err.max():.3e} at x = {xs[err.argmax()]:.4f}   (polynomial form: 4.1e-5)")
print("// scale", repr(float(S)), "offset", NSEG / 2, "clamp hi", repr(float(np.nextafter(np.float32(HI), np.float32(0)))))
packed = (c2s.view(np.uint16).astype(np.uint32) << 16) | c1s.view(np.uint16).astype(np.uint32)     # low half c1, high half c2
print("__device__ const float GELU_TAB_C0[64] = {" + ", ".join(f"{float(v):.9e}f" for v in c0s) + "};")
print("__device__ const unsigned GELU_TAB_C12[64] = {" + ", ".join(f"0x{int(v):08x}u" for v in packed) + "};")
inr = np.abs(xs) <= 4.4
print(f"// in [-4.4, 4.4]: max {err[inr].max():.3e} at x = {xs[inr][err[inr].argmax()]:.4f}; relative to max(|GELU|, 2^-14) worst {np.max(err[inr] / np.maximum(np.abs(ref[inr]), 2.0 ** -14)):.3e}")
