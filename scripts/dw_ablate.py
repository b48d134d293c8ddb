"""Timing ablations of the MFMA depth-wise 7x7 + LayerNorm kernel (gp_dwconv_ln act codes 105 = no conv loop, 106 = no LDS-DMA of the
halo tiles, 108 = neither, 109 = no conv and no global stores; wrong results) at the bench shapes, B = 64 and 128 crops; interleaved rounds, medians."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
for B in (64, 128):
    for (C, H) in ((128, 64), (256, 32), (512, 16)):
        x = torch.randn(B, H, H, C, device="cuda").half()
        w = torch.randn(49, C, device="cuda").half()
        b = torch.randn(C, device="cuda"); lw = torch.randn(C, device="cuda"); lb = torch.randn(C, device="cuda")
        y = torch.empty_like(x)
        res = {}
        for rnd in range(5):
            for act, name in ((0, "full"), (105, "no conv loop"), (106, "no halo DMA"), (108, "no conv, no halo DMA"), (109, "no conv, no global stores")):
                f = lambda: ops.dwconv_ln(x, w, b, lw, lb, y, 7, act=act)
                for _ in range(3): f()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20): f()
                e1.record(); torch.cuda.synchronize()
                res.setdefault(name, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        io = 2 * x.numel() * 2
        print(f"C={C} {H}x{H} B={B}  (in + out {io / 1e6:.0f} MB: {io / 6.3e6:.1f} us at 6.3 TB/s)")
        for name, r in res.items():
            print(f"   {name:24s} median {statistics.median(r):7.1f} us")
