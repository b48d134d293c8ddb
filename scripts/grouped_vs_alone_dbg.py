"""Debug: the crop of a grouped launch whose R differs most from the separate forward -- its rot6d logits, R from both runs and
R recomputed on the host from each run's logits (resnet34_nodcn workload, fp16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from givepose_amd import PoseNet, PoseNetConfig, synth
from givepose_amd.rot_cond import rot6d_amplification, rot_error_bound
cfg = PoseNetConfig(main_backbone="resnet34", use_dcn="")
B = 64
b0, b1 = synth.synth_batch(B, seed=1000), synth.synth_batch(B, seed=1100)       # bench.py's batches of slot 0
both = {k: torch.from_numpy(__import__("numpy").concatenate([b0[k], b1[k]], 0)).cuda() for k in b0}
grouped = PoseNet(cfg, seed=0, dtype=torch.float16, dcn_couple=64).cuda()
alone = PoseNet(cfg, seed=0, dtype=torch.float16).cuda()
og = {k: v.float().cpu().clone() for k, v in grouped.forward_device(both).items() if k in ("rot", "rot6d", "trans", "pred_t", "rot_allo")}
for j, b in enumerate((b0, b1)):
    oa = {k: v.float().cpu().clone() for k, v in alone.forward_device({k: torch.from_numpy(v).cuda() for k, v in b.items()}).items() if k in ("rot", "rot6d", "trans", "pred_t", "rot_allo")}
    sl = slice(j * B, (j + 1) * B)
    dR = (og["rot"][sl] - oa["rot"]).abs().reshape(B, -1).max(1).values
    w = int(dR.argmax())
    print(f"batch {j}: worst crop {w}: |dR| {float(dR[w]):.4f}; bound {float(rot_error_bound(oa['rot6d'], og['rot6d'][sl])[w]):.4f}; amplification {float(rot6d_amplification(oa['rot6d'])[w]):.2f}")
    print("  rot6d alone  ", oa["rot6d"][w].tolist())
    print("  rot6d grouped", og["rot6d"][sl][w].tolist())
    for name in ("rot_allo", "rot", "trans", "pred_t"):
        if name in oa:
            print(f"  {name} alone  ", [round(x, 4) for x in oa[name][w].reshape(-1).tolist()])
            print(f"  {name} grouped", [round(x, 4) for x in og[name][sl][w].reshape(-1).tolist()])
