import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
dev = "cuda"; dt = torch.float16
def bench(M, N, K, variant, splitk, ldx=None, f32out=False):
    x = torch.randn(M, ldx or K, device=dev).to(dt); w = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if f32out else dt); b = torch.randn(N, device=dev)
    f = lambda: ops.gemm(x, w, out, bias=b, epilogue=ops.EPI_LRELU, variant=variant, splitk=splitk, M=M, K=K, ldx=ldx or K)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) * 100)
    return sorted(ts)[2]
for name, (M, N, K, ldx, f32) in {"pnp.fc1": (64, 2048, 8192, None, False), "pnp.fc2": (64, 256, 1024, 2048, True), "red": (4096, 256, 1024, None, False),
                                   "deconv512": (4096, 2304, 512, None, True)}.items():
    row = []
    for v, sk in ((7, 1), (4, 1), (4, 2), (4, 4), (4, 8), (4, 16), (4, 32)):
        try:
            row.append(f"v{v}/k{sk}: {bench(M, N, K, v, sk, ldx, f32):6.1f}")
        except Exception as e:
            row.append(f"v{v}/k{sk}: err")
    print(f"{name:10s} M={M} N={N} K={K}  " + "  ".join(row), " auto:", ops.auto_splitk(M, N, K, 2))
