"""gn_apply_xyz (GroupNorm + GELU + 1x1 out layer) MFMA form against the VALU form (GP_GNXYZ_MFMA=0): timing + error vs fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from givepose_amd import ops
g = torch.Generator().manual_seed(11)
B, R, C = 64, 64, 256
x = torch.randn(B, R * R, C, generator=g).half()
gw, gb = 1 + 0.1 * torch.randn(C, generator=g), 0.1 * torch.randn(C, generator=g)
ow, ob = torch.randn(3, C, generator=g) * C ** -0.5, torch.randn(3, generator=g) * 0.1
xd = x.cuda()
xf = xd.float().view(B, R * R // 64, 64, 32, C // 32)                      # (B, chunk, row, group, channel in group)
part = torch.stack([xf.sum((2, 4)), (xf * xf).sum((2, 4))], -1).contiguous().view(-1)   # (B, HW/64, G, 2): the conv epilogue's layout
nchw, nhwc4 = torch.empty(B, 3, R, R, device="cuda"), torch.empty(B * R * R, 4, device="cuda")
from givepose_amd._lib import ACT_GELU
gwd, gbd, owd, obd = gw.cuda(), gb.cuda(), ow.cuda(), ob.cuda()
f = lambda: ops.groupnorm_apply_xyz(xd, gwd, gbd, owd, obd, nchw, nhwc4, 32, ACT_GELU, part)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): f()
e1.record(); torch.cuda.synchronize()
xr = x[:4].double()
y = F.gelu(F.group_norm(xr.permute(0, 2, 1), 32, gw.double(), gb.double(), 1e-5).permute(0, 2, 1))
ref = (y @ ow.double().t() + ob.double()).permute(0, 2, 1).reshape(4, 3, R, R)
d = (nchw[:4].cpu().double() - ref).abs()
print(f"mfma={os.environ.get('GP_GNXYZ_MFMA', '1')}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us; vs fp64: max abs {d.max():.3e}, mean abs {d.mean():.3e}; nhwc4 == nchw: {torch.equal(nhwc4.view(B, R * R, 4)[..., :3].permute(0, 2, 1).reshape(B, 3, R, R), nchw)}")
i = int(d.flatten().argmax()); bi, ci, pi = i // (3 * R * R), (i // (R * R)) % 3, i % (R * R)
print(f"  worst element: crop {bi} out {ci} pixel {pi}: got {nchw[bi, ci].flatten()[pi].item():.6f} want {ref[bi, ci].flatten()[pi].item():.6f}; row max|pre-act| {y[bi, pi].abs().max().item():.3f}; count > 1e-4: {int((d > 1e-4).sum())}")
