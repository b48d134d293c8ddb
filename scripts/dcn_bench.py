#!/usr/bin/env python3
"""DCNv3 gather at the three MAPEncoder geometries, bs=64 fp16 (GPU box only): us and algorithmic GB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops

B = int(os.environ.get("B", 64))
for R in (64, 32, 16):
    Ho = R // 2
    x = torch.randn(B, R, R, 256, device="cuda").half()
    rows = B * Ho * Ho
    off = (torch.rand(B * R * R, 72, device="cuda") * 2 - 1) * float(os.environ.get("OFF", 3))
    msk = torch.randn(B * R * R, 36, device="cuda")
    om = torch.cat([off, msk], 1).contiguous()          # (rows_full, 108) fp32, as the offset/mask GEMM writes it
    out = torch.empty(B, Ho, Ho, 256, device="cuda", dtype=torch.half)
    f = lambda: ops.dcnv3_forward_into(x, om[:, :72], om[:, 72:], out, 3, 2, 1, 1, 4, 64, 1.0, off_ld=108, mask_ld=108, mask_is_logits=True)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    by = x.numel() * 2 + rows * 108 * 4 + out.numel() * 2
    print(f"R={R}: {us:7.1f} us  {by / us / 1e3:7.0f} GB/s algorithmic")
