"""Probe: ConvNeXt stage-2 fc1 (N = 2048, K = 512, GELU) and fc2 (N = 512, K = 2048, gamma * v + residual) at the row counts of 3 .. 12 crops (M = 256 x crops: the detections of
one frame): the automatic choice against the latency kernel at each tile height (variants 218 / 318 / 418 = 16 / 32 / 64 rows) and the 128 x 128 tile kernels (7; 4 with K ranges)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from givepose_amd import ops
from givepose_amd._lib import GivePoseHipError
for crops in (3, 4, 5, 6, 8, 10, 12):
    M = crops * 256
    for name, (N, K) in (("fc1", (2048, 512)), ("fc2", (512, 2048))):
        x = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        out = torch.zeros(M, N, device="cuda", dtype=torch.half)
        bias = torch.randn(N, device="cuda")
        kw = dict(epilogue=ops.EPI_GELU) if name == "fc1" else dict(epilogue=ops.EPI_SCALE_RES, gamma=torch.randn(N, device="cuda") * 0.1, residual=out)
        arms = {}
        res0 = torch.randn(M, N, device="cuda").half()
        for label, extra in (("auto", {}), ("v18/16", dict(variant=218)), ("v18/32", dict(variant=318)), ("v18/64", dict(variant=418)), ("v7", dict(variant=7)), ("v5", dict(variant=5)), ("v2", dict(variant=2)), ("v8", dict(variant=8)), ("v9", dict(variant=9)), ("v11", dict(variant=11)), ("v12", dict(variant=12)), ("v4 K4", dict(variant=4, splitk=4))):
            try:
                chk = torch.zeros_like(out)
                kw2 = dict(kw, residual=res0) if name == "fc2" else kw
                ops.gemm(x, w, chk, bias=bias, **kw2, **extra); torch.cuda.synchronize(); arms[label] = extra
                if label == "auto": ref = chk.float()
                else: assert float((chk.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max()), (label, float((chk.float() - ref).abs().max()), float(ref.abs().max()))
            except (GivePoseHipError, RuntimeError):
                pass
        t = {a: [] for a in arms}
        for _ in range(5):
            for a, extra in arms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3): ops.gemm(x, w, out, bias=bias, **kw, **extra)
                e0.record()
                for _ in range(20): ops.gemm(x, w, out, bias=bias, **kw, **extra)
                e1.record(); torch.cuda.synchronize()
                t[a].append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"{crops:3d} crops {name} M={M}: " + "  ".join(f"{a} {statistics.median(v):.1f}" for a, v in t.items()), flush=True)
