"""Timing ablations of the weights-in-registers GEMM (variant 16, stage-2 fc1 + GELU) on the investigation build:
   GP_EXTRA_HIPCC_FLAGS=-DGP_WREG_STAMPS GP_BUILD_TAG=stamps python -m givepose_amd.build
   GP_LIB_PATH=givepose_amd/libgivepose_hip_stamps.so python scripts/wreg_ablate.py
Interleaved rounds in one process, medians.  Ablated arms compute wrong results (timing only)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
dev = "cuda"
ARMS = [(16, "product kernel"), (116, "no MFMA"), (216, "no stores"), (416, "no in-loop DMA"), (816, "no GELU"),
        (1016, "no GELU, no stores"), (1416, "no GELU, no stores, no in-loop DMA (MFMA + LDS reads + barrier only)"), (10, "ping-pong 256x256 (v10)")]
for M in (16384, 32768):
    N, K = 2048, 512
    x = torch.randn(M, K).half().to(dev); w = (torch.randn(N, K) * K ** -0.5).half().to(dev); b = torch.randn(N).to(dev)
    out = torch.empty(M, N, dtype=torch.float16, device=dev)
    res = {v: [] for v, _ in ARMS}
    for rnd in range(7):
        for v, _ in ARMS:
            f = lambda: ops.gemm(x, w, out, bias=b, epilogue=ops.EPI_GELU, variant=v, splitk=1)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30): f()
            e1.record(); torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"M={M} N={N} K={K} +GELU   (MFMA floor at 2.5 PFLOP/s: {2.0 * M * N * K / 2.5e15 * 1e6:.1f} us)")
    for v, name in ARMS:
        m = statistics.median(res[v])
        print(f"  {name:75s} median {m:6.1f} us  min {min(res[v]):6.1f}   {2.0 * M * N * K / m / 1e6:5.0f} TF")
