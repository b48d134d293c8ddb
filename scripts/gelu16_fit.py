"""Round 5: GELU for fp16 storage on PACKED fp16 arithmetic (csrc/gemm.hip gelu16_slice, csrc/mlp.hip).
    gelu(x) = max(x, 0) + R(min(|x|, HI)),  R(a) = a (Phi(a) - 1)   (even in x, R -> 0: all errors are absolute)
R as a degree-N polynomial in u = a * 2 / HI - 1 in [-1, 1] (minimax LP fit; the tail |x| > HI is part of the constraint set).
Prints the coefficients (rounded to fp16) and simulates the kernel's arithmetic (every operation rounded to fp16 once, like
v_pk_fma_f16) against the exact GELU: max / rms error over a uniform grid and over x ~ N(0, 1), beside the error of the exact GELU
rounded to fp16 once (what the fp32 polynomial path stores)."""
import numpy as np
from scipy.optimize import linprog
from scipy.special import erf

HI, N = 4.0, 7
Phi = lambda x: 0.5 * (1 + erf(x / np.sqrt(2)))
f16 = lambda v: np.asarray(v, dtype=np.float64).astype(np.float16).astype(np.float64)
fma16 = lambda a, b, c: f16(a * b + c)

a = np.linspace(0, HI, 16001)
u = a * (2 / HI) - 1
B = np.stack([u ** k for k in range(N + 1)], 1)
t = a * (Phi(a) - 1)
one = np.ones((1, N + 1))   # u = 1 (a = HI): the value every |x| > HI gets; R(x) in [R(HI), 0] there
A = np.vstack([np.hstack([B, -np.ones((len(a), 1))]), np.hstack([-B, -np.ones((len(a), 1))]), np.hstack([one, [[-1]]]), np.hstack([-one, [[-1]]])])
b = np.concatenate([t, -t, [0.0], [0.0]])
c = np.zeros(N + 2); c[-1] = 1
r = linprog(c, A_ub=A, b_ub=b, bounds=[(None, None)] * (N + 1) + [(0, None)], method="highs")
coef, e = r.x[:-1], r.x[-1]
c16 = f16(coef)
print("fit error (exact arithmetic): %.3e" % e)
print("coefficients c0..c%d (fp16 values):" % N, ", ".join("%.10ef" % v for v in c16))

def sim(x):
    xh = f16(x)
    aa = np.minimum(np.abs(xh), HI)
    uu = fma16(aa, 2 / HI, -1.0)
    p = fma16(np.full_like(uu, c16[N]), uu, c16[N - 1])
    for k in range(N - 2, -1, -1):
        p = fma16(p, uu, c16[k])
    return f16(np.maximum(xh, 0) + p)

for name, x, w in (("uniform [-6, 6]", np.linspace(-6, 6, 400001), None), ("N(0, 1) weighted", np.linspace(-6, 6, 400001), "g"),
                   ("N(0, 2^2) weighted", np.linspace(-8, 8, 400001), "g2")):
    ref = x * Phi(x)
    wt = np.ones_like(x) if w is None else np.exp(-0.5 * x * x) if w == "g" else np.exp(-0.125 * x * x)
    wt = wt / wt.sum()
    for lab, y in (("packed fp16 path", sim(x)), ("exact GELU -> fp16", f16(ref))):
        d = y - ref
        print(f"{name:20s} {lab:20s} max |err| {np.abs(d).max():.2e}  rms {np.sqrt((wt * d * d).sum()):.2e}  rms rel to rms(gelu) {np.sqrt((wt * d * d).sum() / (wt * ref * ref).sum()):.2e}")
