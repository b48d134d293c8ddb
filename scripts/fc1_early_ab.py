"""Round 6: stage-2 fc1 (gemm_wreg3_kernel, variant 21) alone at 24 / 64 / 128 crops: run once with GP_GEMM_WREG_EARLY=0 (full weight wait) and once without
(the first tile starts while the weight slice streams in), alternating processes: profiles/r06_fc1_early_ab.txt."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
for M in (6144, 16384, 32768):
    N, K = 2048, 512
    x = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    out = torch.zeros(M, N, device="cuda", dtype=torch.half); bias = torch.randn(N, device="cuda")
    ts = []
    for _ in range(7):
        for _ in range(3): ops.gemm(x, w, out, bias=bias, epilogue=ops.EPI_GELU)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): ops.gemm(x, w, out, bias=bias, epilogue=ops.EPI_GELU)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 30 * 1e3)
    print(f"GP_GEMM_WREG_EARLY={os.environ.get('GP_GEMM_WREG_EARLY', '1')} fc1 M={M}: {statistics.median(ts):.2f} us", flush=True)
