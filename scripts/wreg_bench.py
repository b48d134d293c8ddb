"""Variant 16 (weights in registers, K = 512) against the tile kernels: bitwise comparison + timing (GPU box only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
dev = "cuda"
g = torch.Generator().manual_seed(1)
for (M, N, epi) in ((16384, 2048, ops.EPI_GELU), (8192, 2048, ops.EPI_GELU), (16384, 256, ops.EPI_NONE), (4096 + 32, 512, ops.EPI_LRELU), (96, 256, ops.EPI_RELU)):
    K = 512
    x = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
    b = torch.randn(N, generator=g).to(dev)
    ref = torch.empty(M, N, dtype=torch.float16, device=dev)
    ops.gemm(x, w, ref, bias=b, epilogue=epi, variant=7, splitk=1)
    out = torch.full((M, N), 7.0, dtype=torch.float16, device=dev)
    bad = 0
    for it in range(int(os.environ.get("REPS", 20))):
        out.fill_(7.0)
        ops.gemm(x, w, out, bias=b, epilogue=epi, variant=16, splitk=1)
        torch.cuda.synchronize()
        if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
            bad += 1
            if bad == 1:
                d = (out.float() - ref.float()).abs()
                nz = torch.nonzero(d > 0)
                print("  first mismatches:", nz[:6].tolist(), "max", d.max().item(), "count", nz.shape[0])
    line = f"M={M} N={N} epi={epi}: {bad} mismatching runs"
    for var in (16, 0, 8, 10):
        if var in (8, 10) and (M % 256 or N % 256):
            continue
        f = lambda: ops.gemm(x, w, out, bias=b, epilogue=epi, variant=var, splitk=1)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 30 * 1e3
        line += f" | v{var}: {us:.1f} us {2.0 * M * N * K / us / 1e6:.0f} TF"
    print(line, flush=True)
