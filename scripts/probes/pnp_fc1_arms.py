"""Probe: ConvPnPNet fc1 (+ fc1_z side by side: N = 2048, K = 8192, LeakyReLU) at 64 / 128 rows: the automatic choice (128 x 128 tiles, 16 K ranges + reduce kernel) against
other split factors and the latency kernel (variant 18).  33.5 MB of weights: 5.3 us at 6.3 TB/s."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from givepose_amd import ops
from givepose_amd._lib import GivePoseHipError
N, K = 2048, 8192
for M in (128, 64, 32, 16):
    x = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    out = torch.zeros(M, N, device="cuda", dtype=torch.half)
    bias = torch.randn(N, device="cuda")
    arms = {}
    for label, extra in (("auto", {}), ("splitK 4", dict(splitk=4)), ("splitK 8", dict(splitk=8)), ("splitK 16", dict(splitk=16)), ("splitK 32", dict(splitk=32)), ("v18", dict(variant=18)), ("v7", dict(variant=7)), ("v10", dict(variant=10))):
        try:
            ops.gemm(x, w, out, bias=bias, epilogue=ops.EPI_LRELU, **extra)
            torch.cuda.synchronize()
            arms[label] = extra
        except (GivePoseHipError, RuntimeError) as e:
            pass
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
    print(f"M={M:4d}: " + "  ".join(f"{a} {statistics.median(v):.1f}" for a, v in t.items()), flush=True)
