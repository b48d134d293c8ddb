"""Window conv (variant 13) timing at the head shapes, two builds via GP_LIB_PATH; prints a checksum for bitwise comparison."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
g = torch.Generator().manual_seed(5)
for R in (64, 32, 16):
    x = torch.randn(64, R, R, 256, generator=g).half().cuda()
    w = (torch.randn(256, 2304, generator=g) * 0.02).half().cuda()
    out = torch.empty_like(x)
    f = lambda: ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=out, variant=13)
    for _ in range(3): f()
    torch.cuda.synchronize()
    ts = []
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 10 * 1e3)
    ts.sort()
    print(f"R={R}: median {ts[2]:.1f} us  ({2.0 * 64 * R * R * 256 * 2304 / ts[2] / 1e6:.0f} TF)  sha={hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]}")
