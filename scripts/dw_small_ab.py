"""dw7x7 + LN at small batch: MFMA tile kernel against the strip kernel (GP_DW_STRIP_BELOW), C = 512 16x16 and C = 1024 8x8."""
import os, sys, statistics, torch
sys.path.insert(0, ".")
from givepose_amd import ops
for B in (1, 2, 4, 8, 16):
    for (C, H) in ((512, 16), (256, 32), (128, 64)):
        x = torch.randn(B, H, H, C, device="cuda").half()
        wt, bias, lw, lb = torch.randn(49, C, device="cuda").half(), torch.randn(C, device="cuda"), torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        out = torch.empty_like(x)
        ts = []
        for rep in range(6):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.dwconv_ln(x, wt, bias, lw, lb, out, 7)
            e1.record(); torch.cuda.synchronize()
            if rep: ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        print(f"GP_DW_STRIP_BELOW={os.environ.get('GP_DW_STRIP_BELOW', '0')} B={B} C={C} {H}x{H}: {statistics.median(ts):.1f} us", flush=True)
