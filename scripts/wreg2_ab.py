"""Variant 17 (gemm_wreg2_kernel: 16-byte stores, DMA lead 2) against variant 16 (gemm_wreg_kernel): bitwise equality over tile
counts per workgroup 1, 2, 3, 5, ragged, ... and all four epilogues, then interleaved timing (medians over rounds) at the stage-2
fc1 shapes of 64 / 128 crops.  (profiles/r04_wreg2_ab.txt was taken with this script when the kernel still carried its A/B arms:
stagger, priority, prefetch depth, DMA placement.)"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
dev = "cuda"
torch.manual_seed(0)
K = 512
bad = 0
for N in (2048, 1024, 256):
    for M in (32, 1024, 2048, 3072, 5120, 1056, 4128, 16384, 32768, 12288 + 32):
        x = torch.randn(M, K).half().to(dev); w = (torch.randn(N, K) * K ** -0.5).half().to(dev); b = torch.randn(N).to(dev)
        for epi in (ops.EPI_GELU, ops.EPI_NONE, ops.EPI_RELU, ops.EPI_LRELU):
            ref = torch.full((M, N), 7.0, dtype=torch.float16, device=dev)
            ops.gemm(x, w, ref, bias=b, epilogue=epi, variant=16, splitk=1)
            for var in ((17,)):
                for rep in range(3):
                    out = torch.full((M, N), -3.0, dtype=torch.float16, device=dev)
                    ops.gemm(x, w, out, bias=b, epilogue=epi, variant=var, splitk=1)
                    if not torch.equal(out, ref):
                        bad += 1
                        d = (out.float() - ref.float()).abs()
                        print(f"MISMATCH N={N} M={M} epi={epi} v{var} rep{rep}: {int((d > 0).sum())} elements, max {float(d.max()):.3e}, first row {int((d > 0).any(1).nonzero()[0])}")
                        break
print("bitwise comparison: ", "ALL EQUAL" if bad == 0 else f"{bad} MISMATCHES")
ARMS = [(16, "v16 (round 2/3 kernel)"), (17, "v17 (16-byte stores, DMA lead 2)"), (10, "ping-pong 256x256")]
for M in (16384, 32768):
    N = 2048
    x = torch.randn(M, K).half().to(dev); w = (torch.randn(N, K) * K ** -0.5).half().to(dev); b = torch.randn(N).to(dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    res = {v: [] for v, _ in ARMS}
    for rnd in range(9):
        for v, _ in ARMS:
            f = lambda: ops.gemm(x, w, out, bias=b, epilogue=ops.EPI_GELU, variant=v, splitk=1)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): f()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"M={M} N={N} K={K} +GELU")
    for v, name in ARMS:
        m = statistics.median(res[v])
        print(f"  {name:42s} median {m:6.1f} us  min {min(res[v]):6.1f}   {2.0 * M * N * K / m / 1e6:5.0f} TF  = {2.0 * M * N * K / m / 1e6 / 2500:.3f} of peak")
