"""MFMA stem against the VALU stem (GP_STEM_MFMA=0 in a second process): timing, error vs the fp32 formula."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from givepose_amd import ops
g = torch.Generator().manual_seed(7)
B = 64
img = torch.randn(B, 3, 256, 256, generator=g)
w, b = torch.randn(128, 3, 4, 4, generator=g) * 48 ** -0.5, torch.randn(128, generator=g) * 0.1
lw, lb = 1 + 0.1 * torch.randn(128, generator=g), 0.1 * torch.randn(128, generator=g)
out = torch.empty(B, 64, 64, 128, dtype=torch.float16, device="cuda")
a = (img.cuda(), w.reshape(128, 48).t().contiguous().cuda(), b.cuda(), lw.cuda(), lb.cuda(), out)
for _ in range(3): ops.convnext_stem(*a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.convnext_stem(*a)
e1.record(); torch.cuda.synchronize()
ref = F.layer_norm(F.conv2d(img[:8].double(), w.double(), b.double(), stride=4).permute(0, 2, 3, 1), (128,), lw.double(), lb.double(), 1e-6)
d = (out[:8].cpu().double() - ref).abs()
print(f"mfma={os.environ.get('GP_STEM_MFMA', '1')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; vs fp64 formula: max abs {d.max():.3e}, mean abs {d.mean():.3e} (fp16 half-ulp at 1: 4.9e-4)")
