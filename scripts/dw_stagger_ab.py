"""Round 5: dwconv7_ln_mfma_kernel with every second workgroup started n x 640 cycles late (act = 120 + n: timing experiment, results unchanged) -- does taking the two
workgroups of a CU out of lock step overlap one's halo DMA with the other's conv?  Interleaved medians, B = 128.  (The kernel-side switch -- `if (dbg >= 20 && (blockIdx.x & 1))
for (i < dbg - 20) s_sleep(10)` before the first DMA issue -- was removed after the run: profiles/r05_dw_stagger_ab.txt.)"""
import sys, statistics, torch
sys.path.insert(0, ".")
from givepose_amd import ops
B = 128
for C, H in ((512, 16), (256, 32), (128, 64)):
    x = torch.randn(B, H, H, C, device="cuda").half()
    w = torch.randn(49, C, device="cuda").half()
    b = torch.randn(C, device="cuda"); lw = torch.randn(C, device="cuda"); lb = torch.randn(C, device="cuda")
    y = torch.empty_like(x)
    arms = (0, 122, 124, 128, 132, 140)
    t = {a: [] for a in arms}
    for rep in range(9):
        for a in arms:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                ops.dwconv_ln(x, w, b, lw, lb, y, 7, act=a)
            e1.record()
            torch.cuda.synchronize()
            if rep:
                t[a].append(e0.elapsed_time(e1) / 5 * 1e3)
    print(f"C {C} {H}x{H} B {B}: us by delay (x 640 cycles) " + "  ".join(f"{(a - 120 if a else 0):2d}: {statistics.median(v):6.1f}" for a, v in t.items()), flush=True)
