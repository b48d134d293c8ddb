"""Probe: the ConvNeXt stem (conv 4x4 s4 3 -> 128 + LN, stem_mfma_kernel) at 64 / 128 crops of 256 x 256, against its HBM floor (fp32 image in, fp16 NHWC out)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from givepose_amd import ops
g = torch.Generator().manual_seed(7)
w = (torch.randn(48, 128, generator=g) / 7).cuda()
b, lw, lb = (torch.randn(128, generator=g).cuda() for _ in range(3))
for B in (128, 64, 16):
    img = torch.randn(B, 3, 256, 256, generator=g).cuda()
    out = torch.empty(B, 64, 64, 128, dtype=torch.half, device="cuda")
    f = lambda: ops.convnext_stem(img, w, b, lw, lb, out)
    ts = []
    for _ in range(3):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30 * 1e3)
    mb = (img.numel() * 4 + out.numel() * 2) / 1e6
    print(f"B={B}: {mb:.0f} MB in + out ({mb / 6.3:.1f} us at 6.3 TB/s): {min(ts):.1f} us (three measurements {', '.join(f'{t:.1f}' for t in ts)}); checksum {float(out.float().abs().mean()):.6f}", flush=True)
