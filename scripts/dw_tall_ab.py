"""Round 6: dwconv7_ln_tall_kernel (16 x 8 tiles, W = 16) against the 16 x 4 kernel on ConvNeXt stage-2 shapes.
One process times BOTH arms in alternating rounds: the tall kernel forced by act code 110, the older routing by
GP_DW_TALL_MIN=<huge> in the environment of this process (the library reads it once).
Usage: GP_DW_TALL_MIN=1000000000 GP_DW_TALLW_MIN=1000000000 python scripts/dw_tall_ab.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
assert int(os.environ.get("GP_DW_TALL_MIN", "0")) > 10 ** 6 and int(os.environ.get("GP_DW_TALLW_MIN", "0")) > 10 ** 6, "run with GP_DW_TALL_MIN=1000000000 GP_DW_TALLW_MIN=1000000000 so that act=0 is the old routing"
g = torch.Generator().manual_seed(3)
SHAPES = ((512, 16, 16, 128), (512, 16, 16, 64), (512, 16, 16, 256), (128, 64, 64, 128), (128, 64, 64, 64), (256, 32, 32, 128), (256, 32, 32, 64))
if os.environ.get('SMALL') == '1':
    SHAPES = ((512, 16, 16, 64), (512, 16, 16, 48), (512, 16, 16, 32), (512, 16, 16, 24), (512, 16, 16, 16), (512, 16, 16, 96), (512, 16, 16, 80), (256, 32, 32, 32), (128, 64, 64, 8))
for (C, H, Wd, B) in SHAPES:
    x = torch.randn(B, H, Wd, C, generator=g).half().cuda()
    w = (torch.randn(49, C, generator=g) / 7).half().cuda()
    b, lw, lb = (torch.randn(C, generator=g).cuda() for _ in range(3))
    ys = {a: torch.empty_like(x) for a in ((0, 110, 112) if Wd == 16 else (0, 110))}
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
    d = float((ys[0].float() - ys[110].float()).abs().max())
    mb = 2 * x.numel() * 2 / 1e6
    print(f"C={C} {H}x{Wd} B={B} ({mb:.0f} MB in + out: {mb / 6.3:.1f} us at 6.3 TB/s): 16x4 kernel {statistics.median(t[0]):.1f} us | tall {statistics.median(t[110]):.1f} us "
          + (f"| quarter-image tiles {statistics.median(t[112]):.1f} us " if 112 in t else "") + f"| max |tall - old| {d:.2e}", flush=True)
