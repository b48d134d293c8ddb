"""GroupNorm apply + GELU + bilinear x2 of TopDownXyzHead: the two passes (gp_groupnorm_apply, gp_upsample_bilinear2x) against the one-pass
gp_groupnorm_upsample2x (the 16 x 16 / 16 x 8 tile arms existed while GP_GNUP_TOY did: round 4, profiles/r04_gn_upsample_ab.txt), at the two places of the head (16 -> 32 and 32 -> 64), 64 and 128 crops.
Interleaved medians on one device; the three results are compared bit for bit first."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import ops

C, G = 256, 32
g = torch.Generator(device="cuda").manual_seed(0)
for crops in (64, 128):
    for H in (16, 32):
        x = torch.randn(crops, H, H, C, device="cuda", generator=g).half()
        gw, gb = torch.rand(C, device="cuda", generator=g) + 0.5, torch.randn(C, device="cuda", generator=g) * 0.1
        part = torch.zeros(1 << 18, device="cuda")
        chunks = H * H // 64
        xf = x.float().view(crops, chunks, 64, G, C // G)
        part[:crops * chunks * G * 2] = torch.stack([xf.sum((2, 4)), (xf * xf).sum((2, 4))], -1).reshape(-1)
        mid, out2, out1 = torch.empty_like(x), torch.empty(crops, 2 * H, 2 * H, C, dtype=torch.half, device="cuda"), torch.empty(crops, 2 * H, 2 * H, C, dtype=torch.half, device="cuda")

        def two():
            ops.groupnorm(x.view(crops, -1, C), gw, gb, mid.view(crops, -1, C), G, ops.ACT_GELU, part, fused_stats=True)
            ops.upsample_bilinear2x(mid, out2)

        def one():
            ops.groupnorm_upsample2x(x, gw, gb, out1, G, ops.ACT_GELU, part)

        two(); one(); same = torch.equal(out1, out2)
        arms = {"two passes": two, "one pass": one}
        ts = {k: [] for k in arms}
        for rep in range(9):
            for k, f in arms.items():
                f(); torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): f()
                e1.record(); torch.cuda.synchronize()
                if rep: ts[k].append(e0.elapsed_time(e1) / 10 * 1e3)
        alg = crops * H * H * C * 2 * 5 / 1e6      # MB: read the source once, write 4 x
        print(f"{crops} crops {H}x{H} -> {2*H}x{2*H}  bitwise equal: {same};  algorithmic {alg:.0f} MB", flush=True)
        for k, v in ts.items():
            m = statistics.median(v)
            print(f"    {k:24s} {m:7.1f} us   {alg / m:6.2f} TB/s of algorithmic bytes", flush=True)
