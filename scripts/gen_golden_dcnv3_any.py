#!/usr/bin/env python3
"""Golden vectors for the generic DCNv3 forward (fp64, independent h / w geometry, any D) and for the DCNv3 backward,
from the REFERENCE's own `dcnv3_core_pytorch` (network/ops_dcnv3/functions/dcnv3_func.py:172-220) and its autograd -- the
recipe of the reference's test (network/ops_dcnv3/test.py:35-170).  Build container only; writes tests/golden/dcnv3_any_*.npz.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_shim  # noqa: E402

ref_shim.install()
import torch  # noqa: E402
from network.ops_dcnv3.functions.dcnv3_func import dcnv3_core_pytorch  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def case(name, N, H, W, G, D, kh, kw, sh, sw, ph, pw, dh, dw, os_, rc, seed, backward):
    g = torch.Generator().manual_seed(seed)
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    P = kh * kw - rc
    inp = (torch.rand(N, H, W, G * D, generator=g, dtype=torch.float64) - 0.3) * 0.5
    off = (torch.rand(N, Ho, Wo, G * P * 2, generator=g, dtype=torch.float64) - 0.5) * 6
    msk = torch.rand(N, Ho, Wo, G, P, generator=g, dtype=torch.float64) + 1e-5
    msk = (msk / msk.sum(-1, keepdim=True)).reshape(N, Ho, Wo, G * P)
    arrs = dict(input=inp.numpy(), offset=off.numpy(), mask=msk.numpy(),
                params=np.array([kh, kw, sh, sw, ph, pw, dh, dw, G, D, rc]), offset_scale=np.float64(os_))
    if backward:
        inp, off, msk = inp.requires_grad_(), off.requires_grad_(), msk.requires_grad_()
    out = dcnv3_core_pytorch(inp, off, msk, kh, kw, sh, sw, ph, pw, dh, dw, G, D, os_, rc)
    arrs["expected"] = out.detach().numpy()
    if backward:
        go = torch.rand(out.shape, generator=g, dtype=torch.float64) - 0.5
        out.backward(go)
        arrs.update(grad_output=go.numpy(), grad_input=inp.grad.numpy(), grad_offset=off.grad.numpy(), grad_mask=msk.grad.numpy())
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **arrs)
    print("wrote", name, tuple(out.shape))


# the reference test's own forward parameters in double (test.py:17-61), then geometries its square CUDA test never reaches
case("dcnv3_any_fwd_ref", 2, 8, 8, 4, 16, 3, 3, 1, 1, 1, 1, 1, 1, 2.0, 0, 3, False)
case("dcnv3_any_fwd_hw", 2, 9, 12, 3, 6, 3, 5, 2, 1, 2, 2, 1, 1, 1.0, 0, 4, False)   # pad_h == pad_w: the PyTorch core swaps them in F.pad (dcnv3_func.py:186-188)
case("dcnv3_any_fwd_dil_rc", 1, 10, 10, 2, 5, 3, 3, 1, 1, 2, 2, 2, 2, 1.5, 1, 5, False)
# backward: the reference test's channel counts that are not multiples of 4 included (test.py:262-265)
for D in (1, 16, 30):
    case(f"dcnv3_any_bwd_D{D}", 2, 8, 8, 2, D, 3, 3, 1, 1, 1, 1, 1, 1, 2.0, 0, 10 + D, True)
case("dcnv3_any_bwd_hw", 2, 9, 12, 3, 6, 3, 5, 2, 1, 2, 2, 1, 1, 1.0, 0, 44, True)
