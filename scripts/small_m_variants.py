"""GEMM-class shapes of PoseNet at B = 1 (latency_b1): every schedule that accepts the shape, timed as a hipGraph of 24
dependent launches (the launch chain of the real step: no host overhead between them), interleaved medians."""
import os, sys, statistics, torch
os.environ["GP_GEMM_SMALLM"] = "0"       # 'auto' = the tile kernels' choice (+ the caller's split-K); variant 18 explicitly
sys.path.insert(0, ".")
from givepose_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NL = 24
gemms = [("s2 fc1", B * 256, 2048, 512, ops.EPI_GELU), ("s2 fc2", B * 256, 512, 2048, ops.EPI_SCALE_RES),
         ("s3 fc1", B * 64, 4096, 1024, ops.EPI_GELU), ("s3 fc2", B * 64, 1024, 4096, ops.EPI_SCALE_RES)]
convs = [("head conv 16x16", 16), ("head conv 32x32", 32), ("head conv 64x64", 64)]
g = torch.Generator(device="cuda").manual_seed(0)
stream = torch.cuda.Stream()


def graph_time(fn):
    with torch.cuda.stream(stream):
        fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=stream):
            for _ in range(NL):
                fn()
        ts = []
        for rep in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gr.replay()
            e1.record()
            torch.cuda.synchronize()
            if rep:
                ts.append(e0.elapsed_time(e1) / NL * 1e3)
    return statistics.median(ts)


for name, M, N, K, epi in gemms:
    x = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.half)
    res = torch.randn(M, N, device="cuda", generator=g).half()
    gamma, b = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    kw = dict(gamma=gamma, residual=res) if epi == ops.EPI_SCALE_RES else {}
    r = {}
    for v, sk in ((0, None), (7, 1), (4, 1), (5, 1), (9, 1), (8, 1), (2, 1), (16, 1), (17, 1), (218, 1), (318, 1), (418, 1), (4, 2), (4, 4), (4, 8), (4, 16)):
        try:
            r[f"v{v}" + (f" splitK{sk}" if sk and sk > 1 else "") if v else "auto"] = round(graph_time(lambda: ops.gemm(x, w, out, bias=b, epilogue=epi, variant=v, splitk=sk, **kw)), 1)
        except Exception as e:
            r[f"v{v} sk{sk}"] = "n/a"
    print(name, M, N, K, r, flush=True)

for name, R in convs:
    x = torch.randn(B, R, R, 256, device="cuda", generator=g).half()
    w = (torch.randn(256, 9 * 256, device="cuda", generator=g) * (9 * 256) ** -0.5).half()
    out = torch.empty(B, R, R, 256, device="cuda", dtype=torch.half)
    part = torch.zeros(1 << 16, device="cuda")
    r = {}
    for v, sk, gn in ((0, None, True), (0, None, False), (18, 1, True), (218, 1, False), (318, 1, False), (418, 1, False), (7, 1, True), (5, 1, True), (9, 1, True), (4, 1, True), (8, 1, True), (13, 1, True),
                      (4, 2, False), (4, 3, False), (4, 6, False), (4, 9, False), (4, 18, False)):
        key = (f"v{v}" if v else "auto") + (f" splitK{sk}" if sk and sk > 1 else "") + (" +gn" if gn else "")
        try:
            r[key] = round(graph_time(lambda: ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=out, variant=v, gn=(part, 32, R * R) if gn else None) if sk in (None, 1) else
                                      ops.gemm(x, w, out.view(-1, 256), conv=dict(B=B, H=R, W=R, Cin=256, KH=3, KW=3, stride=1, pad=1), variant=v, splitk=sk)), 1)
        except Exception as e:
            r[key] = "n/a"
    print(name, B * R * R, 256, 2304, r, flush=True)
