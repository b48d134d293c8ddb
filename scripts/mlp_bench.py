#!/usr/bin/env python3
"""Fused ConvNeXt MLP (gp_convnext_mlp) against the two-GEMM path at the bs=64 stage shapes (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops

B = int(os.environ.get("B", 64))
dt, dev = torch.float16, "cuda"


def timeit(f, n=10, rounds=3):
    ts = []
    for _ in range(rounds):
        for _ in range(2):
            f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / n * 1e3)
    return sorted(ts)[len(ts) // 2]


for C, hw in ((128, 64), (256, 32)):
    M, HD = B * hw * hw, 4 * C
    x = torch.randn(M, C, device=dev).to(dt)
    res = torch.randn(M, C, device=dev).to(dt)
    w1 = (torch.randn(HD, C, device=dev) * C ** -0.5).to(dt)
    w2 = (torch.randn(C, HD, device=dev) * HD ** -0.5).to(dt)
    b1, b2, gamma = torch.randn(HD, device=dev), torch.randn(C, device=dev), torch.randn(C, device=dev) * 0.1
    w2p = ops.convnext_mlp_pack_w2(w2)
    hid = torch.empty(M, HD, device=dev, dtype=dt)
    o1, o2 = res.clone(), res.clone()

    def two():
        ops.gemm(x, w1, hid, bias=b1, epilogue=ops.EPI_GELU)
        ops.gemm(hid, w2, o1, bias=b2, epilogue=ops.EPI_SCALE_RES, gamma=gamma, residual=o1)

    def fused():
        ops.convnext_mlp(x, w1, b1, w2p, b2, gamma, o2, o2)

    o1.copy_(res); two(); o2.copy_(res); fused()
    err = float((o1.float() - o2.float()).abs().max())
    fl = 4.0 * M * C * HD
    t2, tf = timeit(two), timeit(fused)
    print(f"C={C} M={M}: two-GEMM {t2:7.1f} us ({fl / t2 / 1e6:5.0f} TF)   fused {tf:7.1f} us ({fl / tf / 1e6:5.0f} TF)   max|diff| {err:.3e}")
