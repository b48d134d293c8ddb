"""Interleaved A/B of gp_gemm variants on one shape in one process (medians over rounds).
   M= N= K= EPI=(0 none,1 gelu,4 scale_res) VARS=7,8,10,11 ROUNDS=9"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
dev = "cuda"
M, N, K, EPI = (int(os.environ.get(k, d)) for k, d in (("M", 16384), ("N", 512), ("K", 2048), ("EPI", 4)))
x = torch.randn(M, K).half().to(dev); w = (torch.randn(N, K) * K ** -0.5).half().to(dev); b = torch.randn(N).to(dev)
out = torch.randn(M, N).half().to(dev)
kw = dict(gamma=torch.randn(N).to(dev) * 0.1, residual=out) if EPI == 4 else {}
VARS = [int(v) for v in os.environ.get("VARS", "7,8,10,11").split(",")]   # 4xx = variant 4 with split-K xx
ws = None
res = {v: [] for v in VARS}
for rnd in range(int(os.environ.get("ROUNDS", 9))):
    for var in VARS:
        f = lambda: ops.gemm(x, w, out, bias=b, epilogue=EPI, variant=4 if var >= 400 else var, splitk=var - 400 if var >= 400 else 1, **kw)
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40): f()
        e1.record(); torch.cuda.synchronize()
        res[var].append(e0.elapsed_time(e1) / 40 * 1e3)
print(f"M={M} N={N} K={K} epi={EPI}")
for var in VARS:
    r = res[var]
    print(f"  v{var}: median {statistics.median(r):.1f} us  min {min(r):.1f}  max {max(r):.1f}   ({2.0 * M * N * K / statistics.median(r) / 1e6:.0f} TF)")
