"""Which gemm_big schedule is fastest for the split-operand shapes of PoseNet (bs 64)?  Interleaved medians, one process."""
import sys, statistics, torch
sys.path.insert(0, ".")
from givepose_amd import ops

shapes = [("s2 fc1", 16384, 2048, 512, ops.EPI_GELU), ("s2 fc2", 16384, 512, 2048, ops.EPI_SCALE_RES),
          ("s0 fc1", 262144, 512, 128, ops.EPI_GELU), ("s0 fc2", 262144, 128, 512, ops.EPI_SCALE_RES),
          ("s1 fc1", 65536, 1024, 256, ops.EPI_GELU), ("s1 fc2", 65536, 256, 1024, ops.EPI_SCALE_RES),
          ("s3 fc1", 4096, 4096, 1024, ops.EPI_GELU), ("s3 fc2", 4096, 1024, 4096, ops.EPI_SCALE_RES),
          ("ds2", 16384, 512, 1024, ops.EPI_NONE), ("deconv", 4096, 2304, 1024, ops.EPI_NONE)]
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, epi in shapes:
    x = torch.randn(M, K, device="cuda", generator=g)
    w = ops.split_weights(torch.randn(N, K) * K ** -0.5, "cuda")
    out = torch.empty(M, N, device="cuda")
    res = torch.randn(M, N, device="cuda", generator=g)
    gamma = torch.ones(N, device="cuda")
    b = torch.zeros(N, device="cuda")
    planes = ops.split_planes(x, M, K, K).clone()
    xp = planes.view(torch.float32)[: M * K].view(M, K)     # container with the planes inside
    kw = dict(gamma=gamma, residual=res) if epi == ops.EPI_SCALE_RES else {}
    times = {}
    for v in (7, 8, 10):
        if v in (8, 10) and N % 256 and N % 128:
            continue
        times[v] = []
    for rep in range(7):
        for v in times:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                ops.gemm(xp, w, out, bias=b, epilogue=epi, variant=v, x_planes=True, **kw)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                times[v].append(e0.elapsed_time(e1) / 3 * 1e3)
    print(name, M, N, K, {v: round(statistics.median(t), 1) for v, t in times.items()}, flush=True)
