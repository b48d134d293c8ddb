"""Bitwise fingerprint + timing of the MFMA depth-wise 7x7 + LayerNorm kernels (A/B two builds with GP_LIB_PATH)."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
B = int(os.environ.get("B", 64))
g = torch.Generator().manual_seed(3)
for (C, H) in ((128, 64), (256, 32), (512, 16)):
    x = torch.randn(B, H, H, C, generator=g).half().cuda()
    w = torch.randn(49, C, generator=g).half().cuda()
    b, lw, lb = (torch.randn(C, generator=g).cuda() for _ in range(3))
    y = torch.empty_like(x)
    f = lambda: ops.dwconv_ln(x, w, b, lw, lb, y, 7)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): f()
    e1.record(); torch.cuda.synchronize()
    print(f"C={C} H={H} B={B}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us  sha={hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]}")
    if C == 128:
        st = torch.empty(B * H * H * 2, device="cuda"); y2 = torch.empty_like(x)
        ops.dwconv7_raw_stats(x, w, b, y2, st)
        torch.cuda.synchronize()
        print(f"  raw sha={hashlib.sha1(y2.cpu().numpy().tobytes()).hexdigest()[:12]} stats sha={hashlib.sha1(st.cpu().numpy().tobytes()).hexdigest()[:12]}")
