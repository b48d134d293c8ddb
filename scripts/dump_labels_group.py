"""Kernel labels and event-timed durations of one eager fp16 launch sequence over 2 x 64 crops (the bench's launch shape), sorted by total time:
python scripts/dump_labels_group.py"""
import sys, ctypes, torch
sys.path.insert(0, ".")
from givepose_amd import PoseNet, PoseNetConfig, synth, _lib
lib = _lib.load()
net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, dcn_couple=64).cuda()
B = 128
data = {k: torch.from_numpy(v) for k, v in synth.synth_batch(B, seed=3).items()}
for _ in range(3):
    net.forward_device(data)
torch.cuda.synchronize()
_lib.check(lib.gp_timing_begin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gp_timing_begin")
for _ in range(3):
    net.forward_device(data)
_lib.check(lib.gp_timing_end(), "gp_timing_end")
rows = []
for r in range(500):
    lab = ctypes.create_string_buffer(160)
    c, n, ms, fl, by = ctypes.c_int(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    if lib.gp_timing_top(r, lab, 160, ctypes.byref(c), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) != 0:
        break
    rows.append((ms.value / 3, n.value // 3, c.value, lab.value.decode(), fl.value / 3, by.value / 3))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"sum {tot:.3f} ms per launch sequence of {B} crops")
for ms, n, c, lab, fl, by in rows[:60]:
    print(f"{ms * 1e3:9.1f} us  n={n:3d}  {ms * 1e3 / max(n, 1):8.1f} us each  class {c}  {fl / max(ms, 1e-9) / 1e9:7.0f} TFLOP/s {by / max(ms, 1e-9) / 1e6:7.0f} GB/s  {lab}")
