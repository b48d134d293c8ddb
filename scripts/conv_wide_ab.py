"""3x3 window conv on 512 x 128 tiles (variant 813 = variant 13 with the wide tile forced) against 256 x 256 tiles (variant 913): bitwise
equality of output and fused GroupNorm statistics, then interleaved timing at the head shapes (64 / 128 crops)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
dev = "cuda"
torch.manual_seed(1)
for (B, R) in ((2, 64), (3, 32), (64, 64), (128, 64), (128, 32), (64, 32)):
    C = 256
    x = torch.randn(B, R, R, C, device=dev).half(); w = (torch.randn(C, 9 * C, device=dev) * (9 * C) ** -0.5).half(); b = torch.randn(C, device=dev)
    o13, o813 = torch.zeros(B, R, R, C, dtype=torch.float16, device=dev), torch.ones(B, R, R, C, dtype=torch.float16, device=dev)
    g13, g813 = torch.zeros(B * (R * R // 64) * 32 * 2, device=dev), torch.ones(B * (R * R // 64) * 32 * 2, device=dev)
    for epi in (ops.EPI_NONE, ops.EPI_GELU):
        ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=o13, bias=b, epilogue=epi, variant=913, gn=(g13, 32, R * R))
        ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=o813, bias=b, epilogue=epi, variant=813, gn=(g813, 32, R * R))
        torch.cuda.synchronize()
        print(f"B={B} {R}x{R} epi={epi}: out bitwise {torch.equal(o13, o813)}, GroupNorm statistics bitwise {torch.equal(g13, g813)}", flush=True)
    if B < 64: continue
    res = {913: [], 813: []}
    for rnd in range(7):
        for v in (913, 813):
            f = lambda: ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=o13, bias=b, variant=v, gn=(g13, 32, R * R))
            for _ in range(2): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * B * R * R * C * 9 * C
    for v in (913, 813):
        m = statistics.median(res[v])
        print(f"   {'256 x 256 tiles (v13) ' if v == 913 else '512 x 128 tiles (v813)'} median {m:7.1f} us  min {min(res[v]):7.1f}   {fl / m / 1e6:5.0f} TF = {fl / m / 1e6 / 2500:.3f} of peak")
