"""What GroupNorm-apply + GELU fused into the 3x3 window conv's loader would cost, measured (round-3 review item; DESIGN.md 8.2 had only
estimated it): variant 713 = the window conv with one scale / shift / GELU pass over every window piece in LDS, placed in the load phases
(results are wrong -- dummy constants, no border mask: a LOWER bound of the real thing's cost), against variant 13, next to the
GroupNorm-apply pass it would replace (gp_groupnorm_apply with fused statistics, in place) at the same shape.
Needs the investigation build (the arm spills 28 registers; scratch is banned in the product library):
   GP_EXTRA_HIPCC_FLAGS=-DGP_CONV_GNL GP_BUILD_TAG=gnl python -m givepose_amd.build
   GP_LIB_PATH=$PWD/givepose_amd/libgivepose_hip_gnl.so python scripts/conv_gn_loader_ab.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
dev = "cuda"
def timed(f, n=20, rounds=7):
    ts = []
    for _ in range(rounds):
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return statistics.median(ts)
for (B, R) in ((64, 64), (128, 64), (128, 32)):
    C = 256
    x = torch.randn(B, R, R, C, device=dev).half(); w = (torch.randn(C, 9 * C, device=dev) * (9 * C) ** -0.5).half()
    out = torch.empty(B, R, R, C, dtype=torch.float16, device=dev)
    gnp = torch.zeros(B * (R * R // 64) * 32 * 2, device=dev)
    gw, gb = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    t13 = timed(lambda: ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=out, variant=13, gn=(gnp, 32, R * R)))
    t713 = timed(lambda: ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=out, variant=713, gn=(gnp, 32, R * R)))
    ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=out, variant=13, gn=(gnp, 32, R * R))
    ov = out.view(B, R * R, C)
    tgn = timed(lambda: ops.groupnorm(ov, gw, gb, ov, 32, ops.ACT_GELU, gnp, fused_stats=True))
    print(f"{B} crops {R}x{R}: window conv {t13:7.1f} us | with the normalise pass in its loader {t713:7.1f} us (+{t713 - t13:5.1f}) | the GroupNorm-apply pass it would replace {tgn:6.1f} us"
          f" | net {'+' if t713 - t13 - tgn > 0 else ''}{t713 - t13 - tgn:5.1f} us per conv")
