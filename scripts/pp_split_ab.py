"""Round 5: the ping-pong GEMM (variant 10) with the W pieces of a step's LDS-DMA issued inside the MFMA phase (default) against all four in the load phase
(GP_PP_SPLIT_DMA=0): run this script once per setting, alternating, on one box (the switch is read once per process).  Bitwise check against variant 3
is not possible (other summation order); the values are checked against torch."""
import os, sys, statistics, torch
sys.path.insert(0, ".")
from givepose_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
ops.CO_SCHEDULED = True
CROPS = int(os.environ.get("CROPS", 128))
tag = "split" if os.environ.get("GP_PP_SPLIT_DMA") != "0" else "load-phase"
for name, M, N, K, epi in (("s2 fc2", 256 * CROPS, 512, 2048, ops.EPI_SCALE_RES), ("s3 fc1", 64 * CROPS, 4096, 1024, ops.EPI_GELU), ("s3 fc2", 64 * CROPS, 1024, 4096, ops.EPI_SCALE_RES),
                           ("deconv", 64 * CROPS, 2304, 1024, ops.EPI_NONE), ("ds2", 256 * CROPS, 512, 1024, ops.EPI_NONE), ("ragged", 256 * 37 + 40, 768, 576, ops.EPI_NONE)):
    x = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    out = torch.empty(M, N, dtype=torch.half, device="cuda")
    res = torch.randn(M, N, device="cuda", generator=g).half()
    gamma, b = torch.ones(N, device="cuda"), torch.randn(N, device="cuda", generator=g)
    kw = dict(gamma=gamma, residual=res) if epi == ops.EPI_SCALE_RES else {}
    ops.gemm(x, w, out, bias=b, epilogue=epi, variant=10, **kw)
    rows = slice(0, min(M, 4096))
    lin = x[rows].float() @ w.float().t() + b
    ref = torch.nn.functional.gelu(lin) if epi == ops.EPI_GELU else (res[rows].float() + lin if epi == ops.EPI_SCALE_RES else lin)
    err = (out[rows].float() - ref).abs().max().item() / ref.abs().max().item()
    tail = slice(M - 300, M)
    lin2 = x[tail].float() @ w.float().t() + b
    ref2 = torch.nn.functional.gelu(lin2) if epi == ops.EPI_GELU else (res[tail].float() + lin2 if epi == ops.EPI_SCALE_RES else lin2)
    err2 = (out[tail].float() - ref2).abs().max().item() / ref2.abs().max().item()
    ts = []
    for rep in range(9):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            ops.gemm(x, w, out, bias=b, epilogue=epi, variant=10, **kw)
        e1.record()
        torch.cuda.synchronize()
        if rep:
            ts.append(e0.elapsed_time(e1) / 4 * 1e3)
    t = statistics.median(ts)
    print(f"[{tag:10s}] {name:7s} M{M} N{N} K{K}: {t:7.1f} us {2.0 * M * N * K / t / 1e6:6.0f} TFLOP/s  rel err head {err:.2e} tail {err2:.2e}", flush=True)
