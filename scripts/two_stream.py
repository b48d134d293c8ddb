#!/usr/bin/env python3
"""Throughput with N independent batches in flight (one PoseNet replica + hipGraph + stream each) vs one (GPU box only)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import PoseNet, PoseNetConfig, _lib, synth

B = int(os.environ.get("B", 64))
NS = int(os.environ.get("NS", 2))
dev = torch.device("cuda", 0)
net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True, inflight=NS).to(dev)
for i in range(NS):
    s = net.static_inputs(B, dev, slot=i)
    for k, v in synth.synth_batch(B, seed=1000 + i).items():
        s[k].copy_(torch.from_numpy(v).reshape(s[k].shape))
    for _ in range(3):
        net.forward_device(s, dev, slot=i)
torch.cuda.synchronize()
lib = _lib.load()
STAG = float(os.environ.get("STAGGER_MS", 0))


def run(k_streams, steps=60, stagger=0.0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        if stagger and 0 < i < k_streams:
            time.sleep(stagger * 1e-3)          # phase offset between the slots (first round only)
        plan = net._plan(B, dev, i % k_streams)
        _lib.check(lib.gp_graph_launch(plan["graph"], ctypes.c_void_p(net.stream(i % k_streams).cuda_stream)), "launch")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return B * steps / dt, dt / steps * 1e3


for k in range(1, NS + 1):
    for st in ((0.0, STAG) if STAG and k > 1 else (0.0,)):
        for _ in range(2):
            v, ms = run(k, stagger=st)
        print(f"{k} batch(es) in flight, stagger {st} ms: {v:8.1f} images/s  {ms:.3f} ms/step")
