"""Round 6: stage-3 fc2 (N = 1024, K = 4096) and fc1 (N = 4096, K = 1024) at the row counts of 16 .. 64 crops: automatic choice against forced split-K factors."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
from givepose_amd._lib import GivePoseHipError
for crops in (8, 16, 24, 32, 48, 64):
    M = crops * 64
    for name, (N, K) in (("fc1", (4096, 1024)), ("fc2", (1024, 4096))):
        x = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        out = torch.zeros(M, N, device="cuda", dtype=torch.half)
        bias = torch.randn(N, device="cuda")
        kw = dict(epilogue=ops.EPI_GELU) if name == "fc1" else dict(epilogue=ops.EPI_SCALE_RES, gamma=torch.randn(N, device="cuda") * 0.1, residual=out)
        arms = {}
        for label, extra in (("auto", {}), ("splitK 2", dict(splitk=2)), ("splitK 4", dict(splitk=4)), ("splitK 8", dict(splitk=8)), ("v18", dict(variant=18)), ("v10", dict(variant=10))):
            try:
                ops.gemm(x, w, out, bias=bias, **kw, **extra)
                torch.cuda.synchronize()
                arms[label] = extra
            except (GivePoseHipError, RuntimeError) as e:
                pass
        t = {a: [] for a in arms}
        for _ in range(5):
            for a, extra in arms.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    ops.gemm(x, w, out, bias=bias, **kw, **extra)
                e0.record()
                for _ in range(20):
                    ops.gemm(x, w, out, bias=bias, **kw, **extra)
                e1.record(); torch.cuda.synchronize()
                t[a].append(e0.elapsed_time(e1) / 20 * 1e3)
        print(f"{crops:3d} crops stage-3 {name} M={M}: " + "  ".join(f"{a} {statistics.median(v):.1f}" for a, v in t.items()), flush=True)
