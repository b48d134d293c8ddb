"""Per-label kernel times (eager, hipEvents around every launch) of one ragged multi-frame forward: python scripts/ragged_labels.py 24
(frames of 2-6 detections summing to ~N crops).  All kernel classes, sorted by total time."""
import os, sys, ctypes, torch
sys.path.insert(0, ".")
from givepose_amd import PoseNet, PoseNetConfig, synth, _lib
lib = _lib.load()
net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0).cuda()
for N in [int(a) for a in sys.argv[1:]]:
    sizes, pat = [], [4, 3, 5, 4, 2, 6]
    while sum(sizes) + pat[len(sizes) % 6] <= N:
        sizes.append(pat[len(sizes) % 6])
    data = {k: torch.from_numpy(v) for k, v in synth.synth_batch(sum(sizes), seed=3).items()}
    for plain in (False, True):
        kw = {} if plain else {"groups": sizes}
        net.forward_device(data, **kw)
        torch.cuda.synchronize()
        _lib.check(lib.gp_timing_begin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gp_timing_begin")
        net.forward_device(data, **kw)
        _lib.check(lib.gp_timing_end(), "gp_timing_end")
        rows = []
        for r in range(2000):
            lab = ctypes.create_string_buffer(160)
            c, n, ms, fl, by = ctypes.c_int(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
            if lib.gp_timing_top(r, lab, 160, ctypes.byref(c), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) != 0:
                break
            rows.append((ms.value * 1e3, n.value, lab.value.decode()))
        rows.sort(reverse=True)
        tot = sum(r[0] for r in rows)
        print(f"=== {sum(sizes)} crops, {'ONE coupled batch (plain forward)' if plain else str(len(sizes)) + ' frames (ragged)'}: {sum(r[1] for r in rows)} launches, {tot:.0f} us of kernels")
        for t, n, lab in rows[:28]:
            print(f"   {t:8.1f} us  {n:3d} x {t / max(n, 1):7.1f}  {lab}")
