"""Interleaved A/B of the stage-2 fc1 variants in one process (medians over rounds)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
dev = "cuda"
M, N, K = int(os.environ.get("M", 16384)), 2048, 512
x = torch.randn(M, K).half().to(dev); w = (torch.randn(N, K) * K ** -0.5).half().to(dev); b = torch.randn(N).to(dev)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
VARS = [int(v) for v in os.environ.get("VARS", "16,10,8").split(",")]
res = {v: [] for v in VARS}
for rnd in range(int(os.environ.get("ROUNDS", 9))):
    for var in VARS:
        f = lambda: ops.gemm(x, w, out, bias=b, epilogue=ops.EPI_GELU, variant=var, splitk=1)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): f()
        e1.record(); torch.cuda.synchronize()
        res[var].append(e0.elapsed_time(e1) / 40 * 1e3)
for var in VARS:
    r = res[var]
    print(f"v{var}: median {statistics.median(r):.1f} us  min {min(r):.1f}  max {max(r):.1f}   ({2.0 * M * N * K / statistics.median(r) / 1e6:.0f} TF)")
