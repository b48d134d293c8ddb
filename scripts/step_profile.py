"""Per-kernel table of one eager PoseNet step (hipEvents around every launch, gp_timing_top): label, launches, us, share."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import PoseNet, PoseNetConfig, _lib, synth
B = int(os.environ.get("B", 64)); INF = int(os.environ.get("INFLIGHT", 1))
dt = torch.float32 if os.environ.get("DTYPE") == "f32" else torch.float16
cfg = PoseNetConfig(use_dcn="" if os.environ.get("NODCN") else "dcnv3")
net = PoseNet(cfg, dtype=dt, seed=0, inflight=INF).cuda()
st = net.static_inputs(B, "cuda")
for k, v in synth.synth_batch(B, seed=1000).items():
    st[k].copy_(torch.from_numpy(v).reshape(st[k].shape))
lib = _lib.load()
net.forward_device(st); torch.cuda.synchronize()
reps = 3
lib.gp_timing_begin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
for _ in range(reps):
    net.forward_device(st)
lib.gp_timing_end()
rows, tot = [], 0.0
for r in range(200):
    lab = ctypes.create_string_buffer(160)
    c, n, ms, fl, by = ctypes.c_int(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
    if lib.gp_timing_top(r, lab, 160, ctypes.byref(c), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) != 0:
        break
    rows.append((lab.value.decode(), c.value, n.value // reps, ms.value / reps * 1e3, fl.value / ms.value / 1e9, by.value / ms.value / 1e6))
    tot += ms.value / reps * 1e3
print(f"sum of kernels {tot:.1f} us per step (B={B}, {dt})")
for lab, c, n, us, tf, gbs in rows:
    print(f"{us:8.1f} us {100 * us / tot:5.1f}%  x{n:<3d} {us / n:7.1f} us/launch  {tf:7.1f} TF {gbs:7.0f} GB/s  [{_lib.KC_NAMES[c]}] {lab}")
