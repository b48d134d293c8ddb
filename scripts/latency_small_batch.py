"""Latency of one PoseNet.forward at the batch sizes evaluate.py feeds (the detections of one frame), hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import PoseNet, PoseNetConfig, synth
for dt in (torch.float16, torch.float32):
    net = PoseNet(PoseNetConfig(), dtype=dt, seed=0, use_graph=True).cuda()
    for B in (1, 2, 4, 8, 16, 64):
        st = net.static_inputs(B, "cuda")
        for k, v in synth.synth_batch(B, seed=1).items():
            st[k].copy_(torch.from_numpy(v).reshape(st[k].shape))
        for _ in range(4):
            net.forward_device(st)
        torch.cuda.synchronize()
        n = 30
        t0 = time.perf_counter()
        for _ in range(n):
            net.forward_device(st)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print(f"{str(dt):14s} B={B:3d}: {ms:7.3f} ms per forward = {B / ms * 1e3:8.1f} images/s", flush=True)
