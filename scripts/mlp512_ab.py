"""Round 5: ConvNeXt stage-2 block MLP (C = 512) as ONE launch (gp_convnext_mlp, C = 512: one wave per SIMD, hidden tensor in registers) against the
two-launch path (fc1 on the weights-in-registers kernel + fc2 on the ping-pong tile).  Correctness against torch, then interleaved medians."""
import os, sys, statistics, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
from givepose_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
C, HD = 512, 2048
ops.CO_SCHEDULED = True
def make(M):
    x = torch.randn(M, C, device="cuda", generator=g).half()
    res = torch.randn(M, C, device="cuda", generator=g).half()
    w1 = (torch.randn(HD, C, device="cuda", generator=g) * C ** -0.5).half()
    w2 = (torch.randn(C, HD, device="cuda", generator=g) * HD ** -0.5).half()
    b1, b2, gamma = torch.randn(HD, device="cuda", generator=g), torch.randn(C, device="cuda", generator=g), torch.randn(C, device="cuda", generator=g) * 0.1
    return x, res, w1, w2, b1, b2, gamma
for M in (128, 384, 4096 + 128, 32768):
    x, res, w1, w2, b1, b2, gamma = make(M)
    w2p = ops.convnext_mlp_pack_w2(w2)
    ref = res.float() + gamma * (F.gelu(x.float() @ w1.float().t() + b1) @ w2.float().t() + b2)
    hid = torch.empty(M, HD, device="cuda", dtype=torch.half)
    o1, o2 = res.clone(), res.clone()
    ops.gemm(x, w1, hid, bias=b1, epilogue=ops.EPI_GELU)
    ops.gemm(hid, w2, o1, bias=b2, epilogue=ops.EPI_SCALE_RES, gamma=gamma, residual=o1)
    ops.convnext_mlp(x, w1, b1, w2p, b2, gamma, o2, o2)
    torch.cuda.synchronize()
    e1, e2 = (o1.float() - ref), (o2.float() - ref)
    o3 = res.clone()
    ops.convnext_mlp(x, w1, b1, w2p, b2, gamma, o3, o3)
    print(f"M {M}: two launches max|err| {e1.abs().max().item():.3e} rms {e1.pow(2).mean().sqrt().item():.3e} | fused max|err| {e2.abs().max().item():.3e} rms {e2.pow(2).mean().sqrt().item():.3e} "
          f"(|ref| rms {ref.pow(2).mean().sqrt().item():.2f}); repeat bitwise {torch.equal(o2, o3)}", flush=True)
for CROPS in (64, 128, 256):
    M = 256 * CROPS
    x, res, w1, w2, b1, b2, gamma = make(M)
    w2p = ops.convnext_mlp_pack_w2(w2)
    hid = torch.empty(M, HD, device="cuda", dtype=torch.half)
    o1, o2 = res.clone(), res.clone()
    def two():
        ops.gemm(x, w1, hid, bias=b1, epilogue=ops.EPI_GELU)
        ops.gemm(hid, w2, o1, bias=b2, epilogue=ops.EPI_SCALE_RES, gamma=gamma, residual=o1)
    def fused():
        ops.convnext_mlp(x, w1, b1, w2p, b2, gamma, o2, o2)
    times = {"two launches": [], "fused": []}
    for rep in range(11):
        for name, fn in (("two launches", two), ("fused", fused)):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rep:
                times[name].append(e0.elapsed_time(e1) / 4 * 1e3)
    r = {k: round(statistics.median(t), 1) for k, t in times.items()}
    print(f"stage-2 MLP, {CROPS} crops (M {M}): us {r}  TFLOP/s { {k: round(4.0 * M * C * HD / t / 1e6) for k, t in r.items()} }", flush=True)
