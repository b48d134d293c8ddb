"""Stage-3 ConvNeXt MLP GEMMs (C = 1024: fc1 M x 4096 x 1024 + GELU, fc2 M x 1024 x 4096 + gamma / residual) and the other short-K tile launches of a 128-crop
launch sequence on the tile variants (7: 128x128, two workgroups per CU; 8: 256x128, two per CU; 10: ping-pong 256x256, one per CU): interleaved medians."""
import sys, statistics, torch
sys.path.insert(0, ".")
from givepose_amd import ops
ops.CO_SCHEDULED = False
g = torch.Generator(device="cuda").manual_seed(0)
def run(M, N, K, epi, variants):
    x = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    out = torch.randn(M, N, device="cuda", generator=g).half()
    bias = torch.randn(N, device="cuda", generator=g)
    kw = dict(gamma=torch.randn(N, device="cuda", generator=g) * 0.1, residual=out) if epi == ops.EPI_SCALE_RES else {}
    t = {v: [] for v in variants}
    for rep in range(9):
        for v in variants:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.gemm(x, w, out, bias=bias, epilogue=epi, variant=v, **kw)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                t[v].append(e0.elapsed_time(e1) / 5 * 1e3)
    r = {v: round(statistics.median(a), 1) for v, a in t.items()}
    print(f"M {M} N {N} K {K} epi {epi}: us by variant {r}   TFLOP/s { {v: round(2.0 * M * N * K / u / 1e6) for v, u in r.items()} }", flush=True)
for M in (8192, 4096):
    run(M, 4096, 1024, ops.EPI_GELU, (0, 10, 8, 7))
    run(M, 1024, 4096, ops.EPI_SCALE_RES, (0, 10, 8, 7))
run(8192, 2304, 1024, ops.EPI_NONE, (0, 10, 8, 7))
run(8192, 2304, 512, ops.EPI_NONE, (0, 10, 8, 7))
run(131072, 256, 256, ops.EPI_NONE, (0, 8, 7))
