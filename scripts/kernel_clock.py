"""Round 5: the shader clock the chip holds INSIDE the GEMM-class kernels (MI355X_MICROARCH.md, DVFS give-back item 6).
Needs the investigation build:  GP_EXTRA_HIPCC_FLAGS=-DGP_CLOCK_STAMPS GP_BUILD_TAG=clk python -m givepose_amd.build
and GP_LIB_PATH=givepose_amd/libgivepose_hip_clk.so.  Thread 0 of every workgroup stamps (s_memtime, s_memrealtime) around the main
loop; after ~2 s of back-to-back launches on random data: clock = d memtime / d memrealtime x 100 MHz, median over the workgroups,
and the main loop's share of the launch.  DATA=zeros repeats it on zero-filled operands."""
import os, sys, time, statistics, torch
sys.path.insert(0, ".")
from givepose_amd import ops
g = torch.Generator(device="cuda").manual_seed(0)
ops.CO_SCHEDULED = True
CROPS = int(os.environ.get("CROPS", 128))
ZERO = os.environ.get("DATA") == "zeros"
rnd = (lambda *s: torch.zeros(*s, device="cuda")) if ZERO else (lambda *s: torch.randn(*s, device="cuda", generator=g))

def run(name, launch, flops, nwg):
    st = torch.zeros(nwg * 4 + 64, dtype=torch.int64, device="cuda")
    launch(None)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 0
    while time.time() - t0 < 2.0:
        for _ in range(50):
            launch(None)
        torch.cuda.synchronize()
        n += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        launch(None)
    e1.record()
    launch(st)
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    s = st[: nwg * 4].view(nwg, 4).cpu()
    ok = (s[:, 2] > s[:, 0]) & (s[:, 3] > s[:, 1])
    s = s[ok]
    clk = ((s[:, 2] - s[:, 0]).double() / (s[:, 3] - s[:, 1]).double() * 0.1)      # GHz
    loop_us = (s[:, 3] - s[:, 1]).double() / 100.0
    cyc = (s[:, 2] - s[:, 0]).double()
    ghz = clk.median().item()
    peak = 1024 * 4 * 256 * ghz * 1e9 / 1e12   # fp16 MFMA FLOP/s at that clock: 1024 FLOP / cycle / SIMD
    print(f"{name:44s} {us:7.1f} us  {flops / us / 1e6:6.0f} TFLOP/s | main loop {loop_us.median().item():7.1f} us, {cyc.median().item():9.0f} cycles, clock {ghz:.3f} GHz "
          f"(min {clk.min().item():.3f} max {clk.max().item():.3f}; {int(ok.sum())} workgroups) -> MFMA peak at that clock {peak:5.0f} TFLOP/s, kernel at {flops / us / 1e6 / peak:.2f} of it", flush=True)

K = 512
M, N = 256 * CROPS, 2048
x, w, b = rnd(M, K).half(), (rnd(N, K) * K ** -0.5).half(), rnd(N)
out = torch.empty(M, N, dtype=torch.half, device="cuda")
for v in (17, 19, 20, 21, 22):
    run(f"s2 fc1 v{v} M{M} N{N} K{K} +GELU", lambda st, v=v: ops.gemm(x, w, out, bias=b, epilogue=ops.EPI_GELU, variant=v, splitk=1, _stamps=st), 2.0 * M * N * K, 256)
for name, M, N, K, epi in (("s2 fc2", 256 * CROPS, 512, 2048, ops.EPI_SCALE_RES), ("s3 fc1", 64 * CROPS, 4096, 1024, ops.EPI_GELU), ("gemm 8192^3", 8192, 8192, 8192, ops.EPI_NONE)):
    x, w = rnd(M, K).half(), (rnd(N, K) * K ** -0.5).half()
    out = torch.empty(M, N, dtype=torch.half, device="cuda")
    res, gamma, b = rnd(M, N).half(), torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    kw = dict(gamma=gamma, residual=res) if epi == ops.EPI_SCALE_RES else {}
    nwg = (M // 256) * (N // 256)
    run(f"{name} v10 M{M} N{N} K{K}", lambda st: ops.gemm(x, w, out, bias=b, epilogue=epi, variant=10, splitk=1, _stamps=st, **kw), 2.0 * M * N * K, nwg)
for R in (64, 32):
    conv = dict(B=CROPS, H=R, W=R, Cin=256, KH=3, KW=3, stride=1, pad=1)
    M, N, K = CROPS * R * R, 256, 2304
    x, w = rnd(CROPS, R, R, 256).half(), (rnd(N, K) * K ** -0.5).half()
    out = torch.empty(M, N, dtype=torch.half, device="cuda")
    b = torch.zeros(N, device="cuda")
    run(f"conv3x3 v13 {R}x{R} x{CROPS}", lambda st: ops.gemm(x, w, out, bias=b, variant=13, conv=conv, _stamps=st), 2.0 * M * N * K, M // 512 * 2)
