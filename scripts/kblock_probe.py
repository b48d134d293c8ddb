"""Round-5 probe (the dbg arms it drives -- variants 610 / 710 / 613 / 513 -- were removed from the library after the measurement: profiles/r05_kblock_probe.txt).
What whole-line (K-blocked / channel-planar) operand ADDRESSING buys the LDS-DMA kernels.  Timing only: the dbg arms
read the same buffers in a permuted order (wrong results).  Interleaved medians, one process, one device.
  gemm v10 / 610 (W blocked) / 710 (X and W blocked);  conv v13 / 613 (W blocked) / 513 (W blocked, X planar)."""
import os, sys, statistics, torch
sys.path.insert(0, ".")
from givepose_amd import ops

CROPS = int(os.environ.get("CROPS", 128))
g = torch.Generator(device="cuda").manual_seed(0)
ops.CO_SCHEDULED = True

def bench(fn_by_arm, reps=9, inner=4):
    times = {a: [] for a in fn_by_arm}
    for rep in range(reps):
        for a, fn in fn_by_arm.items():
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                fn()
            e1.record()
            torch.cuda.synchronize()
            if rep:
                times[a].append(e0.elapsed_time(e1) / inner * 1e3)
    return {a: round(statistics.median(t), 1) for a, t in times.items()}

for name, M, N, K, epi in (("s2 fc2", 256 * CROPS, 512, 2048, ops.EPI_SCALE_RES), ("s3 fc1", 64 * CROPS, 4096, 1024, ops.EPI_GELU),
                           ("s3 fc2", 64 * CROPS, 1024, 4096, ops.EPI_SCALE_RES), ("deconv", 64 * CROPS, 2304, 1024, ops.EPI_NONE)):
    x = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.half)
    res = torch.randn(M, N, device="cuda", generator=g).half()
    gamma, b = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    kw = dict(gamma=gamma, residual=res) if epi == ops.EPI_SCALE_RES else {}
    r = bench({v: (lambda v=v: ops.gemm(x, w, out, bias=b, epilogue=epi, variant=v, **kw)) for v in (10, 610, 710)})
    fl = 2.0 * M * N * K
    print(name, M, N, K, r, {a: round(fl / t / 1e6, 0) for a, t in r.items()}, "TFLOP/s", flush=True)

for R in (64, 32, 16):
    conv = dict(B=CROPS, H=R, W=R, Cin=256, KH=3, KW=3, stride=1, pad=1)
    M, N, K = CROPS * R * R, 256, 2304
    x = torch.randn(CROPS, R, R, 256, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.half)
    b = torch.zeros(N, device="cuda")
    r = bench({v: (lambda v=v: ops.gemm(x, w, out, bias=b, variant=v, conv=conv)) for v in (13, 613, 513)})
    fl = 2.0 * M * N * K
    print(f"conv3x3 {R}x{R}", M, r, {a: round(fl / t / 1e6, 0) for a, t in r.items()}, "TFLOP/s", flush=True)
