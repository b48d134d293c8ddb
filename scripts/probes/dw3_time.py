"""Probe: the DCNv3 prefix kernel (dw 3x3 + LN + GELU, C = 256, strip kernel) at the three MAPEncoder geometries of a 64-crop batch (the first 16 crops' pixels),
with and without the GELU, against the HBM floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from givepose_amd import ops
g = torch.Generator().manual_seed(5)
C = 256
for (B, H) in ((64, 64), (64, 32), (64, 16), (128, 64), (128, 32), (128, 16), (32, 64), (16, 64)):
    x = torch.randn(B, H, H, C, generator=g).half().cuda()
    w = (torch.randn(9, C, generator=g) / 3).half().cuda()
    b, lw, lb = (torch.randn(C, generator=g).cuda() for _ in range(3))
    n = B * H * H // 4
    y = torch.empty(n, C, dtype=torch.half, device="cuda")
    res = []
    for act in (ops.ACT_GELU, ops.ACT_NONE, 120 + ops.ACT_GELU, 125 + ops.ACT_GELU):
        f = lambda: ops.dwconv_ln(x, w, b, lw, lb, y, 3, act=act, n_pixels=n)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 30 * 1e3)
    mb = 2 * n * C * 2 / 1e6
    print(f"B={B} {H}x{H}: prefix {n} pixels ({mb:.1f} MB in + out: {mb / 6.3:.1f} us at 6.3 TB/s): routed, with GELU {res[0]:.1f} us | without {res[1]:.1f} us | LDS-tiled kernel forced: 16 x 4 tiles {res[2]:.1f} us, 16 x 2 tiles {res[3]:.1f} us", flush=True)
