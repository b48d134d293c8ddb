"""Microbenchmark of the GroupNorm apply / upsample kernels at the bs=64 head shapes (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops

def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

B, C, G = 64, 256, 32
for HW in (256, 1024, 4096):
    x = torch.randn(B, HW, C, device="cuda").half()
    y = torch.empty_like(x)
    w = torch.randn(C, device="cuda"); b = torch.randn(C, device="cuda")
    part = torch.rand(B, HW // 64, G, 2, device="cuda") * 100 + 2000
    for act, nm in ((ops.ACT_GELU, "gelu"), (ops.ACT_RELU, "relu"), (ops.ACT_NONE, "none")):
        us = t(lambda: ops.groupnorm(x, w, b, y, G, act, part, fused_stats=True))
        print(f"gn_apply HW={HW:5d} {nm}: {us:6.1f} us  ({2 * x.numel() * 2 / us / 1e3:5.0f} GB/s)")
    us = t(lambda: y.copy_(x))
    print(f"torch copy  HW={HW:5d}: {us:6.1f} us  ({2 * x.numel() * 2 / us / 1e3:5.0f} GB/s)")
    if HW == 4096:
        ow = torch.randn(3, C, device="cuda"); ob = torch.randn(3, device="cuda")
        o1 = torch.empty(B, 3, HW, device="cuda"); o2 = torch.empty(B * HW, 4, device="cuda")
        us = t(lambda: ops.groupnorm_apply_xyz(x, w, b, ow, ob, o1, o2, G, ops.ACT_GELU, part))
        print(f"gn_apply_xyz HW={HW}: {us:6.1f} us  ({x.numel() * 2 / us / 1e3:5.0f} GB/s)")
    if HW <= 1024:
        h = int(HW ** 0.5)
        xi = x.view(B, h, h, C); yo = torch.empty(B, 2 * h, 2 * h, C, device="cuda", dtype=torch.half)
        us = t(lambda: ops.upsample_bilinear2x(xi, yo))
        print(f"upsample2x HW={HW}: {us:6.1f} us  ({5 * x.numel() * 2 / us / 1e3:5.0f} GB/s)")
