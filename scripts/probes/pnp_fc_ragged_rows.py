"""Probe: ConvPnPNet's fc GEMMs (M = the crop count) at row counts that are no multiple of 16: the latency kernel (variant 18, which takes such shapes since round 6: partial last
row tile) against the 128 x 128 tile kernels they ran on before (variant 7; variant 4 with 16 K ranges for fc1)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from givepose_amd import ops
for M in (24, 9, 12, 40):
    for name, (N, K) in (("fc1 + fc1_z", (2048, 8192)), ("fc2", (256, 1024))):
        x = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        out = torch.zeros(M, N, device="cuda", dtype=torch.half)
        bias = torch.randn(N, device="cuda")
        arms = {"auto": {}, "v18": dict(variant=18), "v7": dict(variant=7), "v4 splitK 16": dict(variant=4, splitk=16)}
        t = {a: [] for a in arms}
        for _ in range(5):
            for a, extra in arms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    ops.gemm(x, w, out, bias=bias, epilogue=ops.EPI_LRELU, **extra)
                e0.record()
                for _ in range(20):
                    ops.gemm(x, w, out, bias=bias, epilogue=ops.EPI_LRELU, **extra)
                e1.record(); torch.cuda.synchronize()
                t[a].append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"M={M:3d} {name:12s} N={N} K={K}: " + "  ".join(f"{a} {statistics.median(v):.1f} us" for a, v in t.items()), flush=True)
