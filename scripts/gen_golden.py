#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (build container only).

Run:  python scripts/gen_golden.py            (needs /root/reference; never runs on the GPU box)

For every fixture the expected outputs come from the reference's own modules
(network/PoseNet.py, xyz_head.py, conv_pnp_net.py, pose_head.py, ops_dcnv3/..., pose_utils/...)
executed on CPU through scripts/ref_shim.py, with the seeded synthetic weights of
givepose_amd.synth loaded into them.  The script also asserts that oracle/posenet_ref.py
reproduces each of them, so a fixture is never written from an oracle that disagrees with
the reference.  Stored: small inputs + expected outputs; large inputs (roi_img, weights) are
re-derived from their seed and pinned by a checksum.
"""
import json
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_shim  # noqa: E402

FLAGS = ref_shim.install()
import torch  # noqa: E402

from givepose_amd.config import PoseNetConfig  # noqa: E402
from givepose_amd import synth  # noqa: E402
from oracle import posenet_ref as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
torch.set_grad_enabled(False)
SEED = 0


def crc(a):
    return int(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def maxdiff(a, b):
    return float((torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max())


def save(name, **arrs):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"  wrote {name}.npz ({os.path.getsize(path) / 1024:.0f} KB)")


def load_synth_into(module, prefix, seed=SEED, rename=None):
    """Fill a reference module's state_dict with givepose_amd.synth tensors (by canonical name)."""
    sd = module.state_dict()
    new = {}
    for k, v in sd.items():
        canon = rename(k) if rename else k
        if canon is None:
            new[k] = v
            continue
        new[k] = torch.from_numpy(synth.synth_tensor(prefix + canon, tuple(v.shape), seed)).to(v.dtype)
    module.load_state_dict(new, strict=True)
    return module


def gen_dcnv3_core():
    from network.ops_dcnv3.functions.dcnv3_func import dcnv3_core_pytorch
    print("dcnv3 core")
    # (1) the reference's own test parameters (network/ops_dcnv3/test.py:20-33,35-61), stride 1
    torch.manual_seed(3)
    N, H, W, G, D, K = 2, 8, 8, 4, 16, 3
    P = K * K
    inp = torch.rand(N, H, W, G * D) * 0.01
    offset = torch.rand(N, H, W, G * P * 2) * 10
    mask = torch.rand(N, H, W, G, P) + 1e-5
    mask = (mask / mask.sum(-1, keepdim=True)).reshape(N, H, W, G * P)
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        exp = dcnv3_core_pytorch(inp.to(dt), offset.to(dt), mask.to(dt), K, K, 1, 1, 1, 1, 1, 1, G, D, 2.0, 0)
        got = O.dcnv3_forward_ref(inp.to(dt), offset.to(dt), mask.to(dt), K, 1, 1, 1, G, D, 2.0, 0)
        d = maxdiff(exp, got)
        print(f"  s1 {tag}: oracle vs dcnv3_core_pytorch {d:.2e}")
        assert d < (1e-8 if dt == torch.float64 else 1e-6)
        if dt == torch.float32:
            save("dcnv3_s1", input=inp, offset=offset, mask=mask, expected=exp,
                 params=np.array([K, 1, 1, 1, G, D, 0]), offset_scale=2.0)
    # (2) stride 2, PoseNet geometry (G4 x D64), full-resolution offset/mask buffers of which the CUDA
    #     kernel consumes the flat prefix (expected = reference core on that prefix, via the shim op)
    import DCNv3 as ext
    G, D = 4, 64
    for B, H in ((1, 16), (4, 8), (5, 8)):
        r = np.random.Generator(np.random.Philox(key=[SEED, 1000 + B]))
        inp = torch.from_numpy(r.standard_normal((B, H, H, G * D), dtype=np.float32))
        offset = torch.from_numpy(r.uniform(-3, 3, (B, H, H, G * P * 2)).astype(np.float32))
        mask = torch.from_numpy(r.random((B, H, H, G, P), dtype=np.float32) + 1e-3)
        mask = (mask / mask.sum(-1, keepdim=True)).reshape(B, H, H, G * P)
        exp = ext.dcnv3_forward(inp, offset, mask, K, K, 2, 2, 1, 1, 1, 1, G, D, 1.0, 256, 0)
        got = O.dcnv3_forward_ref(inp, offset, mask, K, 2, 1, 1, G, D, 1.0, 0)
        d = maxdiff(exp, got)
        print(f"  s2 B{B}: oracle vs reference core on consumed prefix {d:.2e}")
        assert d < 2e-5
        save(f"dcnv3_s2_B{B}", input=inp, offset=offset, mask=mask, expected=exp,
             params=np.array([K, 2, 1, 1, G, D, 0]), offset_scale=1.0)


def gen_modules():
    from network.xyz_head import TopDownXyzHead
    from network.conv_pnp_net import ConvPnPNet, MAPEncoder
    from network.pose_head import SizeHead
    from network.dcnv3 import DCNv3_C
    from network.pose_utils.rot_reps import rot6d_to_mat_batch
    from network.pose_utils.pose_from_pred_centroid_z import pose_from_pred_centroid_z
    cfg = PoseNetConfig()
    Pn = O.load_params(synth.synth_state_dict(cfg, SEED))
    r = np.random.Generator(np.random.Philox(key=[SEED, 77]))
    rn = lambda *s: torch.from_numpy(r.standard_normal(s, dtype=np.float32))

    print("xyz heads")
    for name, cin in (("xyz_nocs_head", 1024), ("xyz_deform_head", 512)):
        m = load_synth_into(TopDownXyzHead(in_dim=cin, xyz_num_classes=1).eval(), name + ".")
        x = rn(2, cin, 8, 8)
        exp = torch.cat(m(x), dim=1)
        got = O.xyz_head_ref(Pn, x, name + ".")
        d = maxdiff(exp, got)
        print(f"  {name}: {d:.2e}  |out| {float(exp.abs().mean()):.3f}")
        assert d < 2e-5
        save(name, x=x, expected=exp)

    print("size head")
    m = load_synth_into(SizeHead(in_dim=1024, out_dim=3).eval(), "size_head.")
    x = rn(3, 1024, 8, 8)
    exp = m([x])
    d = maxdiff(exp, O.size_head_ref(Pn, x))
    print(f"  {d:.2e}")
    assert d < 1e-5
    save("size_head", x=x, expected=exp)

    print("DCNv3_C module + MAPEncoder")
    m = load_synth_into(DCNv3_C(256, 256, kernel_size=3, stride=2, padding=1, bias=False).eval(),
                        "nocs_encoder.features.3.")
    x = rn(4, 256, 16, 16)
    exp = m(x)
    xc = torch.nn.functional.conv2d(x, Pn["nocs_encoder.features.3.conv.weight"], Pn["nocs_encoder.features.3.conv.bias"])
    got = O.dcnv3_module_ref(Pn, xc.permute(0, 2, 3, 1), "nocs_encoder.features.3.dcnv3.").permute(0, 3, 1, 2)
    d = maxdiff(exp, got)
    print(f"  DCNv3_C: {d:.2e} |out| {float(exp.abs().mean()):.3f}")
    assert d < 2e-5
    save("dcnv3_module", x=x, expected=exp)
    for B in (1, 4, 5):
        m = load_synth_into(MAPEncoder(3, featdim=256).eval(), "nocs_encoder.")
        x = torch.from_numpy(r.uniform(-0.6, 0.6, (B, 3, 64, 64)).astype(np.float32))
        exp = m(x)
        d = maxdiff(exp, O.map_encoder_ref(Pn, x, cfg))
        print(f"  MAPEncoder B{B}: {d:.2e} |out| {float(exp.abs().mean()):.3f}")
        assert d < 5e-5
        save(f"map_encoder_B{B}", x=x, expected=exp)

    print("ConvPnPNet")
    m = load_synth_into(ConvPnPNet(5, featdim=128, rot_dim=6).eval(), "pnp_net.")
    x = torch.from_numpy(r.uniform(-0.8, 0.8, (2, 5, 64, 64)).astype(np.float32))
    rot, t, _ = m(coor_feat=x)
    rot_o, t_o = O.conv_pnp_ref(Pn, x)
    d = max(maxdiff(rot, rot_o), maxdiff(t, t_o))
    print(f"  {d:.2e} rot6d {rot[0].numpy().round(3)}")
    assert d < 1e-5
    save("pnp_net", x=x, rot=rot, t=t)

    print("pose decode")
    for ds in ("CAMERA+Real", "wild6d"):
        B = 6
        d6 = rn(B, 6)
        pt = torch.cat([0.2 * rn(B, 2), 1.0 + 0.3 * torch.rand(B, 1)], 1)
        batch = {k: torch.from_numpy(v) for k, v in synth.synth_batch(B, seed=5).items()}
        Rm = rot6d_to_mat_batch(d6)
        rot, trans = pose_from_pred_centroid_z(Rm, pred_centroids=pt[:, :2], pred_z_vals=pt[:, 2:3],
                                               roi_cams=batch["cam_K"].clone(), roi_centers=batch["bbox_center"],
                                               resize_ratios=batch["resize_ratio"], roi_whs=batch["roi_wh"],
                                               eps=1e-4, is_allo=True, z_type="REL", is_train=False, dataset_name=ds)
        rot_o, trans_o = O.pose_decode_ref(O.rot6d_to_mat_ref(d6), pt, batch["cam_K"], batch["bbox_center"],
                                           batch["resize_ratio"], batch["roi_wh"], ds)
        d = max(maxdiff(rot, rot_o), maxdiff(trans, trans_o), maxdiff(Rm, O.rot6d_to_mat_ref(d6)))
        print(f"  {ds}: {d:.2e}")
        assert d < 1e-6
        save("pose_decode_" + ds.replace("+", "_"), d6=d6, pred_t=pt, rot_allo=Rm, rot=rot, trans=trans,
             **{k: batch[k] for k in ("cam_K", "bbox_center", "resize_ratio", "roi_wh")})


def gen_e2e():
    """The reference PoseNet.forward itself (network/PoseNet.py:173-231); trunk = HF ConvNextModel
    stand-in for timm (ref_shim._HFConvNeXtFeatures)."""
    from network.PoseNet import PoseNet
    print("PoseNet e2e")
    cfg = PoseNetConfig()
    net = PoseNet().eval()

    def rename(k):
        if k.startswith("backbone."):
            t = synth.hf_to_timm(k[len("backbone."):])
            return None if t is None else "backbone." + t
        return k

    load_synth_into(net, "", rename=rename)
    manifest = {k: list(v.shape) for k, v in net.state_dict().items() if not k.startswith("backbone.")}
    ours = synth.param_manifest(cfg)
    assert [k for k in ours if not k.startswith("backbone.")] == list(manifest), "manifest order/name mismatch"
    assert all(tuple(manifest[k]) == tuple(ours[k]) for k in manifest)
    hf_backbone = {("backbone." + synth.hf_to_timm(k[len("backbone."):])): list(v.shape)
                   for k, v in net.state_dict().items()
                   if k.startswith("backbone.") and synth.hf_to_timm(k[len("backbone."):]) is not None}
    assert {k: tuple(v) for k, v in hf_backbone.items()} == {k: tuple(v) for k, v in ours.items() if k.startswith("backbone.")}
    with open(os.path.join(GOLD, "state_dict_manifest.json"), "w") as f:
        json.dump({"non_backbone_from_reference": manifest,
                   "backbone_hf_mapped_to_timm_names_UNVERIFIED_vs_timm": hf_backbone}, f, indent=0)
    Pn = O.load_params(synth.synth_state_dict(cfg, SEED))
    inter = {}
    def hook(keys):
        def fn(mod, args, out):
            for k, v in zip(keys, out if isinstance(out, (tuple, list)) else (out,)):
                inter[k] = v
        return fn

    net.backbone.register_forward_hook(hook(["feat"]))
    net.nocs_encoder.register_forward_hook(hook(["nocs_feat"]))
    net.pnp_net.register_forward_hook(hook(["rot6d", "pred_t"]))
    for B in (1, 4, 5):
        npb = synth.synth_batch(B, seed=100 + B)
        data = {k: torch.from_numpy(v) for k, v in npb.items()}
        out = net(data, "cpu", do_loss=False)
        ref = O.posenet_forward_ref(Pn, data, cfg, return_intermediates=True)
        ds = {k: maxdiff(out[k], ref[k]) for k in ("rot", "trans", "size", "mask", "nocs_coor", "ivfc_coor")}
        ds.update({k: maxdiff(inter[k], ref[k]) for k in inter})
        print(f"  B{B}: " + " ".join(f"{k}={v:.1e}" for k, v in ds.items()))
        print(f"     rot6d[0]={inter['rot6d'][0].numpy().round(3)} t[0]={inter['pred_t'][0].numpy().round(3)} "
              f"|feat|={float(inter['feat'].abs().mean()):.3f} |nocs|={float(out['nocs_coor'].abs().mean()):.3f} "
              f"|ivfc|={float(out['ivfc_coor'].abs().mean()):.3f}")
        assert ds["mask"] == 0.0
        assert max(ds[k] for k in ("rot", "trans", "size")) < 1e-4, ds
        assert max(ds[k] for k in ("nocs_coor", "ivfc_coor")) < 1e-4, ds
        small = {k: npb[k] for k in ("roi_coord_2d", "cam_K", "roi_wh", "bbox_center", "resize_ratio", "mean_size")}
        save(f"posenet_e2e_B{B}", seed=SEED, batch_seed=100 + B, roi_img_crc=crc(npb["roi_img"]),
             roi_mask_crc=crc(npb["roi_mask"]), **small, **{"out_" + k: v for k, v in out.items()},
             **{"mid_" + k: v for k, v in inter.items()})


def gen_att():
    """network/attention_pnp_net.py MAPTransformerEncoer (nocsmap_encoder='att'); timm Block = ref_shim restatement."""
    from network.attention_pnp_net import MAPTransformerEncoer
    print("MAPTransformerEncoer")
    cfg = PoseNetConfig(nocsmap_encoder="att")
    m = load_synth_into(MAPTransformerEncoer().eval(), "nocs_encoder.")
    ours = [k for k in synth.param_manifest(cfg) if k.startswith("nocs_encoder.")]
    assert ours == ["nocs_encoder." + k for k in m.state_dict()], "att manifest order/name mismatch"
    r = np.random.Generator(np.random.Philox(key=[SEED, 13]))
    x = torch.from_numpy(r.uniform(-0.6, 0.6, (3, 3, 64, 64)).astype(np.float32))
    exp = m(x)
    Pn = O.load_params(synth.synth_state_dict(cfg, SEED))
    d = maxdiff(exp, O.map_transformer_ref(Pn, x))
    print(f"  oracle vs reference {d:.2e} |out| {float(exp.abs().mean()):.3f}")
    assert d < 2e-5
    save("map_transformer", x=x, expected=exp)


def gen_resnet():
    """network/resnet.py resnet34 trunk (conv1..layer4 in the order of ResNet.forward :137-147; avgpool/fc dropped:
    SURVEY.md 8a row a14).  The class is pure torch, so it is the reference's own arithmetic."""
    from network.resnet import resnet34
    print("resnet34 trunk")
    cfg = PoseNetConfig(main_backbone="resnet34")
    m = resnet34().eval()
    load_synth_into(m, "backbone.", rename=lambda k: None if k.startswith("fc.") else k)
    r = np.random.Generator(np.random.Philox(key=[SEED, 34]))
    x = torch.from_numpy(r.standard_normal((2, 3, 128, 128), dtype=np.float32))
    y = m.maxpool(m.relu(m.bn1(m.conv1(x))))
    y = m.layer4(m.layer3(m.layer2(m.layer1(y))))
    Pn = O.load_params(synth.synth_state_dict(cfg, SEED))
    d = maxdiff(y, O.resnet34_ref(Pn, x)[0])
    print(f"  oracle vs reference {d:.2e} |out| {float(y.abs().mean()):.3f} max {float(y.abs().max()):.2f}")
    assert d < 1e-4
    save("resnet34_trunk", x=x, expected=y)


if __name__ == "__main__":
    which = sys.argv[1:] or ["core", "modules", "e2e", "resnet", "att"]
    if "resnet" in which:
        gen_resnet()
    if "att" in which:
        gen_att()
    if "core" in which:
        gen_dcnv3_core()
    if "modules" in which:
        gen_modules()
    if "e2e" in which:
        gen_e2e()
    print("done")
