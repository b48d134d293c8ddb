"""Test-side tool (not product, not collected by pytest): how accurate would a split-operand MFMA mode be?

Every dense contraction of the CPU oracle (F.linear / dense F.conv2d / F.conv_transpose2d) is re-run as the split-operand
GEMM the HIP `f16x2` mode computes -- x = x_hi + 2^-S x_lo, w = w_hi + 2^-S w_lo with fp16 planes, products
x_hi w_hi + 2^-S (x_hi w_lo + x_lo w_hi), fp32 accumulate -- and compared, like the plain fp32 oracle, against the oracle in
float64.  Prints max abs error of rot / trans / size / maps for: fp32 oracle, split mode, fp16 operands.

    python tests/precision_split.py [B] [S]
"""
import os
import sys
import types

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from givepose_amd import PoseNetConfig, synth   # noqa: E402
from oracle import posenet_ref as O              # noqa: E402

S = 11


def split(t, s):
    hi = t.half().float()
    lo = ((t - hi) * (2.0 ** s)).half().float()
    return hi, lo


def make_shim(mode, s=S):
    """mode: 'split' | 'f16' (operands rounded to fp16, one product)."""
    shim = types.ModuleType("Fshim")
    shim.__dict__.update(F.__dict__)

    def contract(fn, x, w, *a, **k):
        if mode == "f16":
            return fn(x.half().float(), w.half().float(), *a, **k)
        xh, xl = split(x, s)
        wh, wl = split(w, s)
        bias = a[0] if a else k.get("bias")
        a0 = (None,) + tuple(a[1:]) if a else ()
        k0 = dict(k)
        if "bias" in k0:
            k0["bias"] = None
        cross = fn(xh, wl, *a0, **k0) + fn(xl, wh, *a0, **k0)
        out = cross * (2.0 ** -s) + fn(xh, wh, *a0, **k0)
        if bias is not None:
            shape = [1] * out.dim()
            shape[1 if fn is not F.linear else -1] = -1
            out = out + bias.view(shape)
        return out

    def linear(x, w, bias=None):
        return contract(F.linear, x, w, bias)

    def conv2d(x, w, bias=None, stride=1, padding=0, dilation=1, groups=1):
        if groups != 1 or w.shape[1] < 16:      # depth-wise / tiny-Cin convs stay exact fp32 (VALU / hi-lo kernels in the product)
            return F.conv2d(x, w, bias, stride, padding, dilation, groups)
        return contract(F.conv2d, x, w, bias, stride, padding, dilation, groups)

    def conv_transpose2d(x, w, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
        return contract(F.conv_transpose2d, x, w, bias, stride, padding, output_padding, groups, dilation)

    shim.linear, shim.conv2d, shim.conv_transpose2d = linear, conv2d, conv_transpose2d
    return shim


def run(P, data, cfg, shim=None):
    keep = O.F
    if shim is not None:
        O.F = shim
    try:
        with torch.no_grad():
            return O.posenet_forward_ref(P, data, cfg)
    finally:
        O.F = keep


def main():
    global S
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 11
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    cfg = PoseNetConfig()
    sd = synth.synth_state_dict(cfg, 0)
    P32 = O.load_params(sd)
    data = {k: torch.from_numpy(v) for k, v in synth.synth_batch(B, seed=seed).items()}
    torch.set_num_threads(8)
    P64 = {k: (v.double() if v.is_floating_point() else v) for k, v in P32.items()}
    d64 = {k: (v.double() if v.is_floating_point() else v) for k, v in data.items()}
    truth = run(P64, d64, cfg)
    keys = ("rot", "trans", "size", "nocs_coor", "ivfc_coor")
    for name, shim in (("fp32 oracle", None), (f"split S={S}", make_shim("split", S)), ("fp16 operands", make_shim("f16"))):
        out = run(P32, data, cfg, shim)
        errs = {k: float((out[k].double() - truth[k]).abs().max()) for k in keys}
        per = (out["rot"].double() - truth["rot"]).abs().flatten(1).max(1).values
        print(f"{name:16s} " + "  ".join(f"{k} {v:.2e}" for k, v in errs.items()) + f"  rot median {float(per.median()):.2e}", flush=True)


if __name__ == "__main__":
    main()
