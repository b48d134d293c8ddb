"""Run one dw7x7+LN shape a few times (for rocprofv3 --pmc passes): python dw_one.py C H [act]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
C, H = int(sys.argv[1]), int(sys.argv[2])
act = int(sys.argv[3]) if len(sys.argv) > 3 else 0
B = 64
x = torch.randn(B, H, H, C, device="cuda").half()
w = torch.randn(49, C, device="cuda").half()
b = torch.randn(C, device="cuda"); lw = torch.randn(C, device="cuda"); lb = torch.randn(C, device="cuda")
y = torch.empty_like(x)
for _ in range(6):
    ops.dwconv_ln(x, w, b, lw, lb, y, 7, act=act)
torch.cuda.synchronize()
