"""Kernel labels of one eager fp16 forward at the given batch sizes (which schedule runs where): python scripts/dump_labels.py 4 8 16
(BB=resnet34: the ResNet-34 trunk variant)"""
import os, sys, ctypes, torch
sys.path.insert(0, ".")
from givepose_amd import PoseNet, PoseNetConfig, synth, _lib
lib = _lib.load()
net = PoseNet(PoseNetConfig(main_backbone=os.environ.get("BB", "convnext")), dtype=torch.float16, seed=0).cuda()
for B in [int(a) for a in sys.argv[1:]]:
    data = {k: torch.from_numpy(v) for k, v in synth.synth_batch(B, seed=3).items()}
    net.forward_device(data)
    torch.cuda.synchronize()
    _lib.check(lib.gp_timing_begin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gp_timing_begin")
    net.forward_device(data)
    _lib.check(lib.gp_timing_end(), "gp_timing_end")
    print("B", B)
    for r in range(500):
        lab = ctypes.create_string_buffer(160)
        c, n, ms, fl, by = ctypes.c_int(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        if lib.gp_timing_top(r, lab, 160, ctypes.byref(c), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) != 0:
            break
        if c.value == 0:
            print("   ", n.value, lab.value.decode(), "%.1f us" % (ms.value * 1e3 / max(n.value, 1)))
