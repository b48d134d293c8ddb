import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
B = 64
for (C, H) in ((128, 64), (256, 32), (512, 16), (1024, 8)):
    x = torch.randn(B, H, H, C, device="cuda").half()
    w = torch.randn(49, C, device="cuda").half()
    b = torch.randn(C, device="cuda"); lw = torch.randn(C, device="cuda"); lb = torch.randn(C, device="cuda")
    y = torch.empty_like(x)
    for act in [int(a) for a in os.environ.get("MODES", "0,104,107").split(",")]:
        f = lambda: ops.dwconv_ln(x, w, b, lw, lb, y, 7, act=act)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"C={C} H={H} mode={act}: {us:.1f} us  ({2 * x.numel() * 2 / us / 1e3:.0f} GB/s algorithmic)")
