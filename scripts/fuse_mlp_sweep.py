"""Round 6: from how many crops per forward the fused ConvNeXt MLP of stages 0 / 1 (convnext_mlp_kernel, C = 128 / 256) beats fc1 + fc2 as two GEMMs:
forward latency (hipGraph replay) at B crops with PoseNetConfig.fuse_mlp_min_batch = B (fused) and = 10^6 (two GEMMs), alternating, one box."""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import PoseNet, PoseNetConfig, synth
for B in (4, 8, 12, 16, 24, 32, 48):
    data = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_batch(B, seed=5).items()}
    nets = {"fused": PoseNet(PoseNetConfig(fuse_mlp_min_batch=1), seed=0, use_graph=True).cuda(), "two GEMMs": PoseNet(PoseNetConfig(fuse_mlp_min_batch=10 ** 6), seed=0, use_graph=True).cuda()}
    t = {k: [] for k in nets}
    for k, n in nets.items():
        for _ in range(4):
            n.forward_device(data)
    torch.cuda.synchronize()
    for _ in range(5):
        for k, n in nets.items():
            t0 = time.perf_counter()
            for _ in range(30):
                n.forward_device(data)
            torch.cuda.synchronize()
            t[k].append((time.perf_counter() - t0) / 30 * 1e3)
    print(f"B = {B:3d}: " + "  ".join(f"{k} {statistics.median(v):.3f} ms" for k, v in t.items()), flush=True)
    del nets
    torch.cuda.empty_cache()
