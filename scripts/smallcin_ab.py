"""pnp_conv1 (5 -> 128 channels, 3x3 s2) MFMA against VALU form (GP_SMALLCIN_MFMA=0): timing + error vs fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from givepose_amd import ops
g = torch.Generator().manual_seed(9)
B, R = 64, 64
xyz = torch.randn(B, R, R, 3, generator=g); coord = torch.randn(B, 2, R, R, generator=g)
nhwc4 = torch.cat([xyz, torch.zeros(B, R, R, 1)], -1).reshape(B * R * R, 4).cuda()
w5 = torch.randn(128, 5, 3, 3, generator=g) * 45 ** -0.5
out = torch.empty(B, R // 2, R // 2, 128, dtype=torch.float16, device="cuda")
a = (nhwc4, coord.cuda(), w5.reshape(128, 45).t().contiguous().cuda(), out, B, R)
for _ in range(3): ops.pnp_conv1(*a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): ops.pnp_conv1(*a)
e1.record(); torch.cuda.synchronize()
xin = torch.cat([xyz.permute(0, 3, 1, 2), coord], 1)[:8].double()
ref = F.conv2d(xin, w5.double(), None, stride=2, padding=1).permute(0, 2, 3, 1)
d = (out[:8].cpu().double() - ref).abs()
print(f"mfma={os.environ.get('GP_SMALLCIN_MFMA', '1')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; vs fp64: max abs {d.max():.3e}, mean abs {d.mean():.3e}")
