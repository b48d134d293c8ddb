"""fp16 GEMM shapes of PoseNet at bs 64 (CROPS=128: the grouped launch shape): every schedule that accepts the shape, interleaved
medians (one process, one device)."""
import os, sys, statistics, torch
sys.path.insert(0, ".")
from givepose_amd import ops

shapes = [("s2 fc1", 16384, 2048, 512, ops.EPI_GELU, (16, 17, 10, 8, 11, 7)), ("s2 fc2", 16384, 512, 2048, ops.EPI_SCALE_RES, (7, 10, 11, 8, 12, 3, 2)),
          ("s3 fc1", 4096, 4096, 1024, ops.EPI_GELU, (7, 10, 11, 8)), ("s3 fc2", 4096, 1024, 4096, ops.EPI_SCALE_RES, (7, 10, 11, 8)),
          ("ds2 (as gemm)", 16384, 512, 1024, ops.EPI_NONE, (7, 10, 11, 8)), ("deconv", 4096, 2304, 1024, ops.EPI_NONE, (7, 10, 11, 8)),
          ("dcn fold L1", 262144, 256, 256, ops.EPI_NONE, (7, 10, 11, 8)), ("dcn out L1", 65536, 256, 256, ops.EPI_NONE, (7, 10, 11, 8))]
SC = int(os.environ.get("CROPS", 64)) // 64
shapes = [(n, M * SC, N, K, e, v) for (n, M, N, K, e, v) in shapes]
g = torch.Generator(device="cuda").manual_seed(0)
for co in (0, 1):
    ops.CO_SCHEDULED = bool(co)
    for name, M, N, K, epi, variants in shapes:
        x = torch.randn(M, K, device="cuda", generator=g).half()
        w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
        out = torch.empty(M, N, device="cuda", dtype=torch.half)
        res = torch.randn(M, N, device="cuda", generator=g).half()
        gamma, b = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
        kw = dict(gamma=gamma, residual=res) if epi == ops.EPI_SCALE_RES else {}
        times = {v: [] for v in (0,) + tuple(variants)}
        for rep in range(7):
            for v in times:
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(4):
                    ops.gemm(x, w, out, bias=b, epilogue=epi, variant=v, **kw)
                e1.record()
                torch.cuda.synchronize()
                if rep:
                    times[v].append(e0.elapsed_time(e1) / 4 * 1e3)
        print("co_scheduled" if co else "alone       ", name, M, N, K, {("auto" if v == 0 else v): round(statistics.median(t), 1) for v, t in times.items()}, flush=True)
