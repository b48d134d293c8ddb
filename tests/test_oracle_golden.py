"""CPU: the oracle (oracle/posenet_ref.py, oracle/dcnv3_ref.c) against the golden vectors that
scripts/gen_golden.py produced from the reference's own modules."""
import numpy as np
import pytest
import torch

from givepose_amd import synth
from givepose_amd.config import PoseNetConfig
from oracle import posenet_ref as O
from oracle.dcnv3_c import dcnv3_forward_c

T = torch.from_numpy


@pytest.fixture(scope="module")
def params():
    return O.load_params(synth.synth_state_dict(PoseNetConfig(), 0))


@pytest.mark.parametrize("name", ["dcnv3_s1", "dcnv3_s2_B1", "dcnv3_s2_B4", "dcnv3_s2_B5"])
def test_dcnv3_core(golden, name):
    z = golden(name)
    K, s, p, d, G, D, rc = (int(v) for v in z["params"])
    os_ = float(z["offset_scale"])
    got = O.dcnv3_forward_ref(T(z["input"]), T(z["offset"]), T(z["mask"]), K, s, p, d, G, D, os_, rc).numpy()
    assert np.abs(got - z["expected"]).max() < 5e-6
    got_c = dcnv3_forward_c(z["input"], z["offset"], z["mask"], K, s, p, d, G, D, os_, rc)
    assert np.abs(got_c - z["expected"]).max() < 5e-6
    # python and C restatements follow the same kernel -> agree to rounding
    assert np.abs(got_c - got).max() < 2e-6


def test_dcnv3_quarter_buffer_couples_crops(golden):
    """at stride 2 crop b reads the offset rows of crop b//4 -> changing crop 1's offsets must
    change crop 4's output and only crops 4..7's."""
    z = golden("dcnv3_s2_B5")
    K, s, p, d, G, D, rc = (int(v) for v in z["params"])
    off = z["offset"].copy()
    base = O.dcnv3_forward_ref(T(z["input"]), T(off), T(z["mask"]), K, s, p, d, G, D, 1.0, rc)
    off[1] += 0.5
    pert = O.dcnv3_forward_ref(T(z["input"]), T(off), T(z["mask"]), K, s, p, d, G, D, 1.0, rc)
    changed = [(base[b] - pert[b]).abs().max().item() > 0 for b in range(5)]
    assert changed == [False, False, False, False, True]


@pytest.mark.parametrize("name", ["xyz_nocs_head", "xyz_deform_head"])
def test_xyz_head(golden, params, name):
    z = golden(name)
    got = O.xyz_head_ref(params, T(z["x"]), name + ".").numpy()
    assert np.abs(got - z["expected"]).max() < 2e-5


def test_size_head(golden, params):
    z = golden("size_head")
    assert np.abs(O.size_head_ref(params, T(z["x"])).numpy() - z["expected"]).max() < 1e-5


def test_dcnv3_module(golden, params):
    z = golden("dcnv3_module")
    x = T(z["x"])
    xc = torch.nn.functional.conv2d(x, params["nocs_encoder.features.3.conv.weight"], params["nocs_encoder.features.3.conv.bias"])
    got = O.dcnv3_module_ref(params, xc.permute(0, 2, 3, 1), "nocs_encoder.features.3.dcnv3.").permute(0, 3, 1, 2)
    assert np.abs(got.numpy() - z["expected"]).max() < 2e-5


@pytest.mark.parametrize("B", [1, 4, 5])
def test_map_encoder(golden, params, B):
    z = golden(f"map_encoder_B{B}")
    got = O.map_encoder_ref(params, T(z["x"]), PoseNetConfig()).numpy()
    assert np.abs(got - z["expected"]).max() < 1e-4


def test_pnp_net(golden, params):
    z = golden("pnp_net")
    rot, t = O.conv_pnp_ref(params, T(z["x"]))
    assert np.abs(rot.numpy() - z["rot"]).max() < 1e-5 and np.abs(t.numpy() - z["t"]).max() < 1e-5


@pytest.mark.parametrize("ds", ["CAMERA_Real", "wild6d"])
def test_pose_decode(golden, ds):
    z = golden("pose_decode_" + ds)
    Rm = O.rot6d_to_mat_ref(T(z["d6"]))
    assert np.abs(Rm.numpy() - z["rot_allo"]).max() < 1e-6
    rot, trans = O.pose_decode_ref(Rm, T(z["pred_t"]), T(z["cam_K"]), T(z["bbox_center"]), T(z["resize_ratio"]),
                                   T(z["roi_wh"]), "wild6d" if ds == "wild6d" else "CAMERA+Real")
    assert np.abs(rot.numpy() - z["rot"]).max() < 1e-6 and np.abs(trans.numpy() - z["trans"]).max() < 1e-6


def test_posenet_e2e_B1(golden, params):
    """Whole PoseNet.forward against the reference run (trunk = HF ConvNeXt stand-in for timm)."""
    z = golden("posenet_e2e_B1")
    import zlib
    npb = synth.synth_batch(1, seed=int(z["batch_seed"]))
    assert zlib.crc32(npb["roi_img"].tobytes()) == int(z["roi_img_crc"])
    for k in ("roi_coord_2d", "cam_K", "roi_wh", "bbox_center", "resize_ratio", "mean_size"):
        assert np.array_equal(npb[k], z[k]), k
    out = O.posenet_forward_ref(params, {k: T(v) for k, v in npb.items()}, PoseNetConfig(), return_intermediates=True)
    assert np.array_equal(out["mask"].numpy(), z["out_mask"])          # bit-exact integer-like output
    for k, tol in (("rot", 1e-4), ("trans", 1e-4), ("size", 1e-4), ("nocs_coor", 1e-4), ("ivfc_coor", 1e-4)):
        assert np.abs(out[k].numpy() - z["out_" + k]).max() < tol, k
    assert np.abs(out["feat"].numpy() - z["mid_feat"]).max() < 1e-4


def test_resnet34_trunk(golden):
    """a14: the oracle trunk vs the reference's own resnet34 class (network/resnet.py) on seeded weights."""
    z = golden("resnet34_trunk")
    P = O.load_params(synth.synth_state_dict(PoseNetConfig(main_backbone="resnet34"), 0))
    got = O.resnet34_ref(P, T(z["x"]))[0].numpy()
    assert np.abs(got - z["expected"]).max() < 1e-4


def test_map_transformer(golden):
    """a13: oracle MAPTransformerEncoer vs the reference class (timm Block restated from memory in ref_shim: unpinned vs timm)."""
    z = golden("map_transformer")
    P = O.load_params(synth.synth_state_dict(PoseNetConfig(nocsmap_encoder="att"), 0))
    assert np.abs(O.map_transformer_ref(P, T(z["x"])).numpy() - z["expected"]).max() < 2e-5


def _geom(z):
    kh, kw, sh, sw, ph, pw, dh, dw, G, D, rc = (int(v) for v in z["params"])
    return (kh, kw, sh, sw, ph, pw, dh, dw, G, D, float(z["offset_scale"]), rc)


@pytest.mark.parametrize("name", ["dcnv3_any_fwd_ref", "dcnv3_any_fwd_hw", "dcnv3_any_fwd_dil_rc"])
def test_dcnv3_any_forward_oracle(golden, name):
    """Generic-geometry fp64 forward (C restatement of cuh:216-282) vs the reference's dcnv3_core_pytorch."""
    from oracle.dcnv3_c import dcnv3_forward_any_c
    z = golden(name)
    got = dcnv3_forward_any_c(z["input"], z["offset"], z["mask"], *_geom(z))
    assert np.abs(got - z["expected"]).max() < 1e-6       # dcnv3_core_pytorch builds its grid from fp32 linspace


@pytest.mark.parametrize("name", ["dcnv3_any_bwd_D1", "dcnv3_any_bwd_D16", "dcnv3_any_bwd_D30", "dcnv3_any_bwd_hw"])
def test_dcnv3_any_backward_oracle(golden, name):
    """fp64 backward (C restatement of cuh:386-487 / :82-140) vs autograd through the reference's dcnv3_core_pytorch."""
    from oracle.dcnv3_c import dcnv3_backward_any_c
    z = golden(name)
    gi, go, gm = dcnv3_backward_any_c(z["input"], z["offset"], z["mask"], z["grad_output"], *_geom(z))
    for got, key in ((gi, "grad_input"), (go, "grad_offset"), (gm, "grad_mask")):
        assert np.abs(got - z[key]).max() < 1e-6 * max(1.0, np.abs(z[key]).max()), key
