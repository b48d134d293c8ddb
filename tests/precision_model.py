"""Test-side tool (not product, not collected by pytest): where does the fp16-storage error of the path come from?

Re-runs the CPU oracle with fp16 ROUNDING injected at the sites where the HIP fp16 mode stores a tensor (or feeds an
MFMA operand), under several policies, and prints the max abs error of rot / trans / size / maps against the fp32
oracle.  Used in round 2 to decide which buffers to keep in fp32.

    python tests/precision_model.py [B]
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from givepose_amd import PoseNetConfig, synth   # noqa: E402
from oracle import posenet_ref as O              # noqa: E402


def q(t):
    return t.half().float()


class Policy:
    def __init__(self, res=True, branch=True, headconv=True, gnout=True, w=True, encoder=True, pnp=True):
        self.res, self.branch, self.headconv, self.gnout, self.w, self.encoder, self.pnp = res, branch, headconv, gnout, w, encoder, pnp

    def Q(self, on, t):
        return q(t) if on else t


def convnext(P, img, cfg, pol):
    g = lambda k: P["backbone." + k]
    W = lambda k: pol.Q(pol.w, g(k))
    x = F.conv2d(img, g("stem_0.weight"), g("stem_0.bias"), stride=4)            # stem runs on fp32 VALU
    x = O._ln_cl(x.permute(0, 2, 3, 1), g("stem_1.weight"), g("stem_1.bias"), 1e-6).permute(0, 3, 1, 2)
    x = pol.Q(pol.res, x)
    for s, n in enumerate(cfg.convnext_depths):
        if s > 0:
            t = O._ln_cl(x.permute(0, 2, 3, 1), g(f"stages_{s}.downsample.0.weight"), g(f"stages_{s}.downsample.0.bias"), 1e-6).permute(0, 3, 1, 2)
            t = pol.Q(pol.branch, t)
            x = F.conv2d(t, W(f"stages_{s}.downsample.1.weight"), g(f"stages_{s}.downsample.1.bias"), stride=2)
            x = pol.Q(pol.res, x)
        for b in range(n):
            p = f"stages_{s}.blocks.{b}."
            xin = pol.Q(pol.branch, x)                                           # dw conv reads an fp16 copy
            y = F.conv2d(xin, W(p + "conv_dw.weight"), g(p + "conv_dw.bias"), padding=3, groups=x.shape[1])
            y = O._ln_cl(y.permute(0, 2, 3, 1), g(p + "norm.weight"), g(p + "norm.bias"), 1e-6)
            y = pol.Q(pol.branch, y)
            y = F.gelu(F.linear(y, W(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias")))
            y = pol.Q(pol.branch, y)
            y = F.linear(y, W(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
            x = pol.Q(pol.res, x + (g(p + "gamma") * y).permute(0, 3, 1, 2))
    return x


def xyz_head(P, x, prefix, pol):
    g = lambda k: P[prefix + k]
    W = lambda k: pol.Q(pol.w, g(k))
    x = F.conv_transpose2d(pol.Q(pol.gnout, x), W("features.0.weight"), None, stride=2, padding=1, output_padding=1)
    x = pol.Q(pol.headconv, x)
    x = pol.Q(pol.gnout, F.gelu(O._gn(x, g("features.1.weight"), g("features.1.bias"))))
    for i in (3, 4, 6, 7, 9, 10):
        if i in (6, 9):
            x = pol.Q(pol.gnout, F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True))
        x = pol.Q(pol.headconv, F.conv2d(x, W(f"features.{i}.conv.weight"), None, padding=1))
        x = F.gelu(O._gn(x, g(f"features.{i}.norm.weight"), g(f"features.{i}.norm.bias")))
        if i != 10:
            x = pol.Q(pol.gnout, x)
    return F.conv2d(x, g("out_layer.weight"), g("out_layer.bias"))


def run(P, data, cfg, pol):
    f = lambda k: data[k].float()
    feat = convnext(P, f("roi_img"), cfg, pol)
    size = O.size_head_ref(P, feat)
    nocs = xyz_head(P, feat, "xyz_nocs_head.", pol)
    if pol.encoder:       # MAPEncoder operands in fp16 (weights + activations)
        Pq = {k: (q(v) if (k.startswith("nocs_encoder.") and k.endswith("weight") and v.dim() > 1) else v) for k, v in P.items()}
        nf = q(O.map_encoder_ref(Pq, nocs, cfg))
    else:
        nf = O.map_encoder_ref(P, nocs, cfg)
    red = F.conv2d(pol.Q(pol.gnout, feat), pol.Q(pol.w, P["feat_reducer.weight"]), P["feat_reducer.bias"])
    ivfc = xyz_head(P, torch.cat([pol.Q(pol.gnout, red), nf], 1), "xyz_deform_head.", pol)
    Pp = {k: (q(v) if (pol.pnp and k.startswith("pnp_net.") and k.endswith("weight") and v.dim() > 1 and not k.startswith(("pnp_net.fc_", "pnp_net.features.0"))) else v)
          for k, v in P.items()}
    rot6d, pred_t = O.conv_pnp_ref(Pp, torch.cat([ivfc, f("roi_coord_2d")], 1))
    ms = f("mean_size")
    size = size + ms / ms.norm(dim=1).unsqueeze(-1)
    rot, trans = O.pose_decode_ref(O.rot6d_to_mat_ref(rot6d), pred_t, f("cam_K"), f("bbox_center"), f("resize_ratio"), f("roi_wh"), cfg.dataset, cfg.t_type)
    return {"rot": rot, "trans": trans, "size": size, "nocs_coor": nocs, "ivfc_coor": ivfc, "feat": feat, "rot6d": rot6d}


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    cfg = PoseNetConfig()
    P = O.load_params(synth.synth_state_dict(cfg, 0))
    data = {k: torch.from_numpy(v) for k, v in synth.synth_batch(B, seed=7).items()}
    torch.set_num_threads(8)
    with torch.no_grad():
        ref = run(P, data, cfg, Policy(False, False, False, False, False, False, False))
        pols = {
            "A all fp16 storage (round 1)": Policy(),
            "B fp32 residual stream": Policy(res=False),
            "C B + fp32 head conv outputs (GN inputs)": Policy(res=False, headconv=False),
            "D only weights rounded": Policy(False, False, False, False, True, False, False),
            "E only trunk branch operands + weights": Policy(res=False, branch=True, headconv=False, gnout=False, w=True, encoder=False, pnp=False),
            "F A but fp32 trunk entirely": Policy(res=False, branch=False),
        }
        print(f"B={B}; |rot6d| ~ {float(ref['rot6d'].abs().mean()):.3e}; |feat| ~ {float(ref['feat'].abs().mean()):.3e}")
        for name, pol in pols.items():
            out = run(P, data, cfg, pol)
            errs = {k: float((out[k] - ref[k]).abs().max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor", "feat")}
            print(f"{name:45s} " + "  ".join(f"{k} {v:.2e}" for k, v in errs.items()))


if __name__ == "__main__":
    main()
