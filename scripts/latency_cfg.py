"""B = 1 / 4 hipGraph-replay latency of a PoseNet configuration, three measurements of 50 replays each (bench.py's latency leg, repeated):
BB=resnet34 python scripts/latency_cfg.py"""
import os, sys, time, torch
sys.path.insert(0, ".")
from givepose_amd import PoseNet, PoseNetConfig, synth
cfg = PoseNetConfig(main_backbone=os.environ.get("BB", "convnext"), use_dcn=os.environ.get("DCN", "dcnv3"))
net = PoseNet(cfg, dtype=torch.float16, seed=0, use_graph=True, inflight=1).cuda()
for B in (1, 4):
    few = {k: torch.from_numpy(v).cuda() for k, v in synth.synth_batch(B, seed=5).items()}
    for _ in range(4):
        net.forward_device(few)
    torch.cuda.synchronize()
    ms = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(50):
            net.forward_device(few)
        torch.cuda.synchronize()
        ms.append(round((time.perf_counter() - t0) / 50 * 1e3, 3))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        net.forward_device(few)
    e1.record()
    torch.cuda.synchronize()
    print(f"{cfg.main_backbone} B={B}: host wall {ms} ms per forward; device events {e0.elapsed_time(e1) / 50:.3f} ms", flush=True)
