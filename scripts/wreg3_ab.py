"""Round 5: stage-2 fc1 on the weights-in-registers kernels: variant 17 (16x16x32 MFMAs, fp32 GELU slices), 19 (32x32x16 MFMAs, the same GELU),
20 (32x32x16, GELU on packed fp16).  Correctness against torch (fp32 formula on the fp16-rounded operands) and interleaved medians."""
import os, sys, statistics, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from givepose_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
K = 512
for (M, N, epi) in [(2048, 1024, ops.EPI_GELU), (64, 256, ops.EPI_GELU), (32, 256, ops.EPI_GELU), (8192 + 64, 256, ops.EPI_GELU), (4096 + 32, 768, ops.EPI_LRELU), (96, 256, ops.EPI_NONE), (33 * 32, 256, ops.EPI_RELU), (16384, 2048, ops.EPI_GELU)]:
    x = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    b = torch.randn(N, device="cuda", generator=g)
    lin = x.float() @ w.float().t() + b
    ref = {ops.EPI_NONE: lin, ops.EPI_GELU: F.gelu(lin), ops.EPI_RELU: F.relu(lin), ops.EPI_LRELU: F.leaky_relu(lin, 0.1)}[epi]
    for v in (17, 19, 20, 21, 22):
        if v in (20, 21) and epi != ops.EPI_GELU:
            continue
        out = torch.full((M, N), 7.0, dtype=torch.half, device="cuda")
        ops.gemm(x, w, out, bias=b, epilogue=epi, variant=v, splitk=1)
        d = (out.float() - ref)
        print(f"M{M} N{N} epi{epi} v{v}: max|err| {d.abs().max().item():.3e} rms {d.pow(2).mean().sqrt().item():.3e} rel {(d.norm() / ref.norm()).item():.3e}", flush=True)
ops.CO_SCHEDULED = True
for CROPS in (64, 128):
    M, N = 256 * CROPS, 2048
    x = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    b = torch.randn(N, device="cuda", generator=g)
    out = torch.empty(M, N, dtype=torch.half, device="cuda")
    times = {v: [] for v in (17, 19, 20, 21, 22)}
    for rep in range(11):
        for v in times:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                ops.gemm(x, w, out, bias=b, epilogue=ops.EPI_GELU, variant=v, splitk=1)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                times[v].append(e0.elapsed_time(e1) / 4 * 1e3)
    r = {v: round(statistics.median(t), 1) for v, t in times.items()}
    print(f"fc1 M{M} N{N} K{K} +GELU: us", r, "TFLOP/s", {v: round(2.0 * M * N * K / t / 1e6) for v, t in r.items()}, flush=True)
