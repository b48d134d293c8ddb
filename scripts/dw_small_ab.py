"""dw7x7 + LN at small batch (the detections of one frame), one arm per process (the switches are read once):
   GP_DW_MFMA_MIN=0                          the MFMA tile kernel wherever it applies (round 3 / 4 default at every batch)
   GP_DW_MFMA_MIN=1000000 GP_DW_NARROW_BELOW=0        strip kernel, 8 pixels per thread
   GP_DW_MFMA_MIN=1000000 GP_DW_NARROW_BELOW=1000000  strip kernel, 2 pixels per thread
Timed as a hipGraph of 24 dependent launches (in place ping-pong between two buffers), us per launch."""
import os, sys, statistics, torch
sys.path.insert(0, ".")
from givepose_amd import ops
NL = 24
arm = f"MFMA_MIN={os.environ.get('GP_DW_MFMA_MIN', 'default')} NARROW_BELOW={os.environ.get('GP_DW_NARROW_BELOW', 'default')}"
stream = torch.cuda.Stream()
for B in (1, 2, 4, 8, 16):
    row = {}
    for (C, H) in ((128, 64), (256, 32), (512, 16), (1024, 8)):
        x = torch.randn(B, H, H, C, device="cuda").half()
        wt, bias, lw, lb = (torch.randn(49, C, device="cuda") * 0.1).half(), torch.randn(C, device="cuda"), torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        out = torch.empty_like(x)
        with torch.cuda.stream(stream):
            ops.dwconv_ln(x, wt, bias, lw, lb, out, 7)
            torch.cuda.synchronize()
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=stream):
                for i in range(NL):
                    ops.dwconv_ln(x if i % 2 == 0 else out, wt, bias, lw, lb, out if i % 2 == 0 else x, 7)
            ts = []
            for rep in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
                if rep: ts.append(e0.elapsed_time(e1) / NL * 1e3)
        row[f"C{C} {H}x{H}"] = round(statistics.median(ts), 1)
    print(f"[{arm}] B={B}: {row}", flush=True)
