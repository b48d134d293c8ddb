"""Run one GEMM shape/variant a few times (for rocprofv3 --pmc passes):
python gemm_one.py M N K variant            plain GEMM
python gemm_one.py conv R variant [epi]     3x3 256->256 conv at RxR, 64 crops
python gemm_one.py pw1|pw2 variant          stage-2 ConvNeXt fc1 (GELU) / fc2 (scale + residual)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
a = sys.argv[1:]
kw, conv = {}, None
if a[0] == "conv":
    R, var = int(a[1]), int(a[2])
    conv = dict(B=64, H=R, W=R, Cin=256, KH=3, KW=3, stride=1, pad=1)
    M, N, K = 64 * R * R, 256, 2304
    x = torch.randn(64, R, R, 256, device="cuda").half()
elif a[0] in ("pw1", "pw2"):
    var = int(a[1])
    M, N, K = (16384, 2048, 512) if a[0] == "pw1" else (16384, 512, 2048)
    x = torch.randn(M, K, device="cuda").half()
else:
    M, N, K, var = [int(v) for v in a[:4]]
    x = torch.randn(M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
out = torch.empty(M, N, device="cuda", dtype=torch.half)
bias = torch.randn(N, device="cuda")
epi = ops.EPI_NONE
if a[0] == "pw1":
    epi = ops.EPI_GELU
if a[0] == "pw2":
    epi, kw = ops.EPI_SCALE_RES, dict(gamma=torch.randn(N, device="cuda"), residual=out)
for _ in range(6):
    ops.gemm(x, w, out, bias=bias, epilogue=epi, variant=var, conv=conv, **kw)
torch.cuda.synchronize()
