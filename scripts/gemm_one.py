"""Run one GEMM shape/variant a few times (for rocprofv3 --pmc passes): python gemm_one.py M N K variant [conv]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
M, N, K, var = [int(v) for v in sys.argv[1:5]]
x = torch.randn(M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
out = torch.empty(M, N, device="cuda", dtype=torch.half)
bias = torch.randn(N, device="cuda")
for _ in range(6):
    ops.gemm(x, w, out, bias=bias, epilogue=ops.EPI_NONE, variant=var)
torch.cuda.synchronize()
