"""One PoseNet.forward at B crops (default 1: the detections of one frame, evaluation/evaluate.py:89-117) replayed as a hipGraph, for
`rocprofv3 --kernel-trace`: scripts/trace_summary.py then gives the per-kernel durations INSIDE the graph and the wall of one replay
(wall - sum of kernels = what the launch gaps cost).  Prints the back-to-back latency too."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import PoseNet, PoseNetConfig, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True).cuda()
st = net.static_inputs(B, "cuda")
for k, v in synth.synth_batch(B, seed=1).items():
    st[k].copy_(torch.from_numpy(v).reshape(st[k].shape))
for _ in range(4):
    net.forward_device(st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    net.forward_device(st)
torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / n * 1e3:.3f} ms per forward (hipGraph replay, back to back)", flush=True)
