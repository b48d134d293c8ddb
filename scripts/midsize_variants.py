"""Round 6: which gp_gemm variant is fastest for ConvNeXt stage-2 fc1 / fc2 at the row counts of 16 .. 96 crops (the ragged multi-frame launches and the
strictly serial bs-64 forward): every variant that accepts the shape, interleaved medians of hipGraph-free launch trains on one box.
python scripts/midsize_variants.py"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops
from givepose_amd._lib import GivePoseHipError
VARS = [0, 4, 7, 8, 10, 12, 16, 17, 18, 21]
for crops in (16, 24, 32, 48, 64, 96):
    M = crops * 256
    for name, (N, K) in (("fc1", (2048, 512)), ("fc2", (512, 2048))):
        x = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
        out = torch.zeros(M, N, device="cuda", dtype=torch.half)
        bias = torch.randn(N, device="cuda")
        kw = dict(epilogue=ops.EPI_GELU) if name == "fc1" else dict(epilogue=ops.EPI_SCALE_RES, gamma=torch.randn(N, device="cuda") * 0.1, residual=out)
        ok = []
        for v in VARS:
            try:
                ops.gemm(x, w, out, bias=bias, variant=v, **kw)
                torch.cuda.synchronize()
                ok.append(v)
            except (GivePoseHipError, RuntimeError):
                pass
        t = {v: [] for v in ok}
        for _ in range(5):
            for v in ok:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    ops.gemm(x, w, out, bias=bias, variant=v, **kw)
                e0.record()
                for _ in range(20):
                    ops.gemm(x, w, out, bias=bias, variant=v, **kw)
                e1.record(); torch.cuda.synchronize()
                t[v].append(e0.elapsed_time(e1) / 20 * 1e3)
        med = {v: statistics.median(t[v]) for v in ok}
        best = min((m, v) for v, m in med.items() if v != 0)
        print(f"{crops:3d} crops {name} M={M}: " + "  ".join(f"v{v} {med[v]:.1f}" for v in ok) + f"  | automatic (v0) {med[0]:.1f} us, best forced v{best[1]} {best[0]:.1f} us", flush=True)

# ---- the heads' 3x3 256 -> 256 convs (with fused GroupNorm statistics) at mid-size crop counts
print("3x3 conv 256 -> 256 + GroupNorm statistics:")
for crops in (16, 24, 32, 48, 64):
    for R in (16, 32, 64):
        x = torch.randn(crops, R, R, 256, device="cuda").half()
        w = (torch.randn(256, 2304, device="cuda") * 2304 ** -0.5).half()
        out = torch.zeros(crops, R, R, 256, device="cuda", dtype=torch.half)
        part = torch.zeros(crops * (R * R // 16) * 64, device="cuda")
        ok = []
        for v in (0, 7, 8, 10, 13, 18):
            try:
                ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=out, variant=v, gn=(part, 32, R * R, 64))
                torch.cuda.synchronize()
                ok.append(v)
            except (GivePoseHipError, RuntimeError):
                pass
        t = {v: [] for v in ok}
        for _ in range(5):
            for v in ok:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(3):
                    ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=out, variant=v, gn=(part, 32, R * R, 64))
                e0.record()
                for _ in range(10):
                    ops.conv2d_nhwc(x, w, 3, 3, 1, 1, out=out, variant=v, gn=(part, 32, R * R, 64))
                e1.record(); torch.cuda.synchronize()
                t[v].append(e0.elapsed_time(e1) / 10 * 1e3)
        med = {v: statistics.median(t[v]) for v in ok}
        print(f"{crops:3d} crops {R}x{R}: " + "  ".join(f"v{v} {med[v]:.1f}" for v in ok), flush=True)
