#!/usr/bin/env python3
"""Per-shape microbenchmark of gp_gemm on the shapes PoseNet launches at bs=64 (GPU box only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops

B = int(os.environ.get("B", 64))
VARS = [int(v) for v in os.environ.get("VARS", "0").split(",")]
ROUNDS = int(os.environ.get("ROUNDS", 3))
dt = torch.float16
dev = "cuda"
shapes = []
for s, (d, h) in enumerate(((128, 64), (256, 32), (512, 16), (1024, 8))):
    M = B * h * h
    shapes.append((f"s{s}.pw1", dict(M=M, N=4 * d, K=d, epi=ops.EPI_GELU)))
    shapes.append((f"s{s}.pw2", dict(M=M, N=d, K=4 * d, epi=ops.EPI_SCALE_RES)))
    if s > 0:
        shapes.append((f"ds{s}", dict(conv=dict(B=B, H=2 * h, W=2 * h, Cin=d // 2, KH=2, KW=2, stride=2, pad=0), N=d)))
for r in (16, 32, 64):
    shapes.append((f"head.conv3x3@{r}", dict(conv=dict(B=B, H=r, W=r, Cin=256, KH=3, KW=3, stride=1, pad=1), N=256)))
shapes.append(("deconv1024", dict(M=B * 64, N=2304, K=1024, f32out=True)))
shapes.append(("enc.proj@64", dict(M=B * 4096, N=256, K=256)))
shapes.append(("enc.om@64", dict(M=B * 1024, N=108, K=256, f32out=True)))
shapes.append(("pnp.conv@32", dict(conv=dict(B=B, H=32, W=32, Cin=128, KH=3, KW=3, stride=2, pad=1), N=128)))
shapes.append(("fc1", dict(M=B, N=2048, K=8192, epi=ops.EPI_LRELU)))
shapes.append(("red", dict(M=B * 64, N=256, K=1024)))
for k in (64, 512, 2048):
    for nm, e in (("none", ops.EPI_NONE), ("gelu", ops.EPI_GELU), ("relu", ops.EPI_RELU)):
        shapes.append((f"abl.k{k}.{nm}", dict(M=16384, N=2048, K=k, epi=e)))
shapes.append(("abl.k512.f32out", dict(M=16384, N=2048, K=512, f32out=True)))
shapes.append(("square4k", dict(M=4096, N=4096, K=4096)))
shapes.append(("square8k", dict(M=8192, N=8192, K=8192)))

def run(name, s):
    conv = s.get("conv")
    N = s["N"]
    if conv:
        Ho = (conv["H"] + 2 * conv["pad"] - conv["KH"]) // conv["stride"] + 1
        M = conv["B"] * Ho * Ho
        K = conv["KH"] * conv["KW"] * conv["Cin"]
        x = torch.randn(conv["B"], conv["H"], conv["W"], conv["Cin"], device=dev).to(dt)
    else:
        M, K = s["M"], s["K"]
        x = torch.randn(M, K, device=dev).to(dt)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if s.get("f32out") else dt)
    bias = torch.randn(N, device=dev)
    epi = s.get("epi", ops.EPI_NONE)
    kw = {}
    if epi == ops.EPI_SCALE_RES:
        kw = dict(gamma=torch.randn(N, device=dev), residual=out)
    fl = 2.0 * M * N * K
    res = {}
    for rnd in range(ROUNDS):                      # interleaved rounds in one process (A/B arms share the device state)
        for var in VARS:
            f = lambda: ops.gemm(x, w, out, bias=bias, epilogue=epi, conv=conv, variant=var, **kw)
            for _ in range(2):
                f()
            n = 10
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                f()
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(var, []).append(e0.elapsed_time(e1) / n * 1e3)
    txt = "  ".join(f"v{var}: {sorted(t)[len(t)//2]:7.1f}us {fl / sorted(t)[len(t)//2] / 1e6:6.0f}TF" for var, t in res.items())
    print(f"{name:18s} M={M:7d} N={N:5d} K={K:5d}  {txt}")

only = sys.argv[1:] 
for name, s in shapes:
    if only and not any(o in name for o in only):
        continue
    run(name, s)
