"""Small-batch latency A/B (hipGraph replay, back to back): PoseNetConfig switches that trade throughput for a shorter
launch chain.  usage: python scripts/latency_ab.py [B ...]"""
import sys, time, dataclasses, torch
sys.path.insert(0, ".")
from givepose_amd import PoseNet, PoseNetConfig, synth

dev = torch.device("cuda:0")
Bs = [int(a) for a in sys.argv[1:]] or [1, 4, 16]
arms = [("default", {}), ("fused MLP (stages 0-1) at any batch", dict(fuse_mlp_min_batch=1)), ("plain fc1 / fc2", dict(fuse_mlp=False)), ("defer_ln", dict(defer_ln=True))]
for B in Bs:
    one = {k: torch.from_numpy(v).to(dev) for k, v in synth.synth_batch(B, seed=1000).items()}
    res = {}
    ref = None
    for name, kw in arms:
        cfg = dataclasses.replace(PoseNetConfig(), **kw)
        net = PoseNet(cfg, dtype=torch.float16, use_graph=True, seed=0)
        for _ in range(5):
            out = net.forward_device(one, dev)
        torch.cuda.synchronize(dev)
        ts = []
        for rep in range(5):
            t0 = time.perf_counter()
            for _ in range(50):
                net.forward_device(one, dev)
            torch.cuda.synchronize(dev)
            ts.append((time.perf_counter() - t0) / 50 * 1e3)
        rot = net._plans[(B, 0, False)]["buf"]["rot6d"].float().cpu().clone()     # the 6-D logits: R itself is ill-conditioned for some crops
        if ref is None:
            ref = rot
        res[name] = (round(sorted(ts)[len(ts) // 2], 3), float((rot - ref).abs().max()))
        del net
    print(f"B={B}", {k: f"{v[0]} ms (rot6d vs arm 0 {v[1]:.1e})" for k, v in res.items()}, flush=True)
