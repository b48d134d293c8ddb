"""Round 6: the pair-tile form of dwconv7_ln_tall_kernel (two 8 x 8 images per 16-column tile; ConvNeXt stage 3, C = 1024) against the strip
kernel that served stage 3 until now.  One process, alternating rounds: act code 113 forces the pair form, 107 the strip kernel.
Usage: python scripts/dw_pair_ab.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
g = torch.Generator().manual_seed(3)
C, H = 1024, 8
for B in (128, 96, 64, 48, 32, 24, 16, 8, 4, 2):
    x = torch.randn(B, H, H, C, generator=g).half().cuda()
    w = (torch.randn(49, C, generator=g) / 7).half().cuda()
    b, lw, lb = (torch.randn(C, generator=g).cuda() for _ in range(3))
    ys = {a: torch.empty_like(x) for a in (107, 113, 114)}
    def timed(a, n=40):
        f = lambda: ops.dwconv_ln(x, w, b, lw, lb, ys[a], 7, act=a)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t = {a: [] for a in ys}
    for _ in range(5):
        for a in ys: t[a].append(timed(a))
    d = float((ys[107].float() - ys[113].float()).abs().max())
    mb = 2 * x.numel() * 2 / 1e6
    print(f"C={C} {H}x{H} B={B} ({mb:.1f} MB in + out: {mb / 6.3:.1f} us at 6.3 TB/s): strip kernel {statistics.median(t[107]):.1f} us | pair tiles, 4 rows {statistics.median(t[113]):.1f} us, 2 rows {statistics.median(t[114]):.1f} us "
          f"| max |pair - strip| {d:.2e}", flush=True)
