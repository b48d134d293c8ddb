#!/usr/bin/env python3
"""Throughput with N independent batches in flight (one PoseNet replica + hipGraph + stream each) vs one (GPU box only)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import PoseNet, PoseNetConfig, _lib, synth

B = int(os.environ.get("B", 64))
NS = int(os.environ.get("NS", 2))
dev = torch.device("cuda", 0)
nets = [PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True).to(dev) for _ in range(NS)]
stat = []
for i, n in enumerate(nets):
    s = n.static_inputs(B, dev)
    for k, v in synth.synth_batch(B, seed=1000 + i).items():
        s[k].copy_(torch.from_numpy(v).reshape(s[k].shape))
    stat.append(s)
    for _ in range(3):
        n.forward_device(s, dev)
torch.cuda.synchronize()
lib = _lib.load()


def run(k_streams, steps=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        n = nets[i % k_streams]
        plan = n._plan(B, dev)
        _lib.check(lib.gp_graph_launch(plan["graph"], ctypes.c_void_p(n._stream.cuda_stream)), "launch")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return B * steps / dt, dt / steps * 1e3


for k in range(1, NS + 1):
    for _ in range(2):
        v, ms = run(k)
    print(f"{k} batch(es) in flight: {v:8.1f} images/s  {ms:.3f} ms/step")
