"""GPU parity, whole path: givepose_amd.PoseNet (HIP kernels through the C ABI) against
  (a) golden vectors captured from the reference's own PoseNet.forward (tests/golden/posenet_e2e_B*.npz,
      trunk = HF ConvNeXt stand-in for timm), and
  (b) the oracle run on the GPU box's CPU for a batch composition the fixtures do not hold.

Tolerances (north_star: 1e-4 abs on R/t/s):
  fp32 storage + fp32 MFMA accumulate : R, t, s <= 1e-4 abs; coordinate maps <= 2e-4 abs; mask bit-exact.
  fp16 storage (throughput mode)      : R <= 3e-2 (B <= 5) / 6e-2 (worst crop of 64), t/s <= 3e-2 relative-to-scale,
      maps <= 2e-2 abs -- fp16 operand rounding (weights alone: 1.5e-3 on R); reported, not claimed to meet 1e-4.
"""
import numpy as np
import re

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model(dtype, **kw):
    from givepose_amd import PoseNet, PoseNetConfig
    return PoseNet(PoseNetConfig(**kw), dtype=dtype, seed=0).cuda()


def _batch(B, seed):
    from givepose_amd import synth
    return {k: torch.from_numpy(v) for k, v in synth.synth_batch(B, seed=seed).items()}


def _launch_labels(net, data):
    """{kernel label: launches} of one eager forward (gp_timing_top), to assert WHICH kernels a configuration runs."""
    import ctypes
    from givepose_amd import _lib
    lib = _lib.load()
    net.forward_device(data)
    torch.cuda.synchronize()
    _lib.check(lib.gp_timing_begin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gp_timing_begin")
    net.forward_device(data)
    _lib.check(lib.gp_timing_end(), "gp_timing_end")
    out = {}
    for r in range(500):
        lab = ctypes.create_string_buffer(160)
        c, n, ms, fl, by = ctypes.c_int(), ctypes.c_long(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        if lib.gp_timing_top(r, lab, 160, ctypes.byref(c), ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(by)) != 0:
            break
        out[lab.value.decode()] = n.value
    return out


@pytest.fixture(scope="module")
def net32():
    return _model(torch.float32)


@pytest.fixture(scope="module")
def net16():
    return _model(torch.float16)


@pytest.mark.parametrize("B", [1, 4, 5])
def test_fp32_matches_reference_golden(golden, net32, B):
    z = golden(f"posenet_e2e_B{B}")
    data = _batch(B, int(z["batch_seed"]))
    dev = net32.forward_device(data)
    mid = {k: dev[k].float().cpu().numpy() for k in ("rot6d", "pred_t", "feat")}
    feat = mid["feat"].transpose(0, 3, 1, 2)
    print("feat", np.abs(feat - z["mid_feat"]).max(), "rot6d", np.abs(mid["rot6d"] - z["mid_rot6d"]).max(),
          "pred_t", np.abs(mid["pred_t"] - z["mid_pred_t"]).max())
    out = net32(data, "cuda")
    assert out["rot"].device.type == "cpu" and out["trans"].device.type == "cuda"    # reference contract
    assert np.array_equal(out["mask"].cpu().numpy(), z["out_mask"])                   # bit-exact
    err = {k: float(np.abs(out[k].cpu().numpy() - z["out_" + k]).max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    print(B, err)
    assert np.abs(feat - z["mid_feat"]).max() < 2e-4
    assert np.abs(mid["rot6d"] - z["mid_rot6d"]).max() < 1e-4 and np.abs(mid["pred_t"] - z["mid_pred_t"]).max() < 1e-4
    assert err["nocs_coor"] < 2e-4 and err["ivfc_coor"] < 2e-4
    assert err["rot"] < 1e-4 and err["trans"] < 1e-4 and err["size"] < 1e-4


def test_fp32_matches_oracle_other_batch(net32):
    """B=8: crops 4..7 read crop 1's offset rows -- checked against the oracle on this host."""
    from givepose_amd import synth
    from givepose_amd.config import PoseNetConfig
    from oracle import posenet_ref as O
    data = _batch(8, 321)
    P = O.load_params(synth.synth_state_dict(PoseNetConfig(), 0))
    ref = O.posenet_forward_ref(P, data, PoseNetConfig())
    out = net32(data, "cuda")
    for k, tol in (("rot", 1e-4), ("trans", 1e-4), ("size", 1e-4), ("nocs_coor", 2e-4), ("ivfc_coor", 2e-4)):
        assert float((out[k].cpu() - ref[k]).abs().max()) < tol, k
    assert torch.equal(out["mask"].cpu(), ref["mask"])


@pytest.mark.parametrize("B", [2, 3, 6, 7, 9])
def test_fp32_matches_oracle_batch_sweep(net32, B):
    """Batch sizes evaluate.py really feeds (the detections of one frame): every remainder of B mod 4 moves the DCNv3
    quarter-buffer coupling (crop b reads the offset rows of crop b // 4) and the npre prefix."""
    from givepose_amd import synth
    from givepose_amd.config import PoseNetConfig
    from oracle import posenet_ref as O
    data = _batch(B, 900 + B)
    ref = O.posenet_forward_ref(O.load_params(synth.synth_state_dict(PoseNetConfig(), 0)), data, PoseNetConfig())
    out = net32(data, "cuda")
    for k, tol in (("rot", 1e-4), ("trans", 1e-4), ("size", 1e-4), ("nocs_coor", 2e-4), ("ivfc_coor", 2e-4)):
        assert float((out[k].cpu() - ref[k]).abs().max()) < tol, (B, k)
    assert torch.equal(out["mask"].cpu(), ref["mask"])


def test_batch_above_im2col_step_must_be_a_multiple_like_the_reference():
    """dcnv3_cuda.cu:46-49: batch % min(batch, im2col_step = 256) == 0 or the op refuses (257 crops -> error, not garbage)."""
    from givepose_amd import _lib, ops
    x = torch.zeros(257, 4, 4, 256, device="cuda", dtype=torch.float16)
    off = torch.zeros(257, 4, 4, 72, device="cuda", dtype=torch.float16)
    msk = torch.zeros(257, 4, 4, 36, device="cuda", dtype=torch.float16)
    with pytest.raises(_lib.GivePoseHipError, match="must divide im2col_step"):
        ops.dcnv3_forward(x, off, msk, 3, 3, 2, 2, 1, 1, 1, 1, 4, 64, 1.0, 256)


@pytest.fixture(scope="module")
def oracle64():
    """The oracle on the bench shape (64 crops; crop b reads the offset rows of crop b // 4, npre = nq + r + 8 rows of the
    prefix are computed), once per wiring: {use_dcn: (data, reference outputs)}."""
    from givepose_amd import synth
    from givepose_amd.config import PoseNetConfig
    from oracle import posenet_ref as O
    cache = {}

    def get(use_dcn):
        if use_dcn not in cache:
            cfg = PoseNetConfig(use_dcn=use_dcn)
            data = _batch(64, 640)
            torch.set_num_threads(min(16, torch.get_num_threads()))
            cache[use_dcn] = (data, O.posenet_forward_ref(O.load_params(synth.synth_state_dict(cfg, 0)), data, cfg, return_intermediates=True))
        return cache[use_dcn]
    return get


@pytest.mark.parametrize("use_dcn", ["dcnv3", ""])
def test_fp32_bs64_matches_oracle(oracle64, use_dcn):
    """BASELINE configs 2 and 1 at the bench batch: fp32 storage meets north_star's 1e-4 on R / t / s."""
    data, ref = oracle64(use_dcn)
    out = _model(torch.float32, use_dcn=use_dcn)(data, "cuda")
    err = {k: float((out[k].cpu() - ref[k]).abs().max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    print("fp32 bs64", repr(use_dcn), err)
    assert torch.equal(out["mask"].cpu(), ref["mask"])
    assert err["rot"] < 1e-4 and err["trans"] < 1e-4 and err["size"] < 1e-4
    assert err["nocs_coor"] < 2e-4 and err["ivfc_coor"] < 2e-4


@pytest.mark.parametrize("use_dcn", ["dcnv3", ""])
def test_fp16_bs64_close_to_oracle(oracle64, use_dcn):
    """The benchmarked mode at the benchmarked shape (hipGraph replay, as bench.py runs it) against the oracle."""
    from givepose_amd import PoseNet, PoseNetConfig
    data, ref = oracle64(use_dcn)
    net = PoseNet(PoseNetConfig(use_dcn=use_dcn), dtype=torch.float16, seed=0, use_graph=True).cuda()
    for _ in range(3):
        out = net(data, "cuda")
    dev_out = net.forward_device(data)
    rot6d = dev_out["rot6d"].float().cpu()
    err = {k: float((out[k].cpu() - ref[k]).abs().max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    print("fp16 bs64", repr(use_dcn), err)
    assert torch.equal(out["mask"].cpu(), ref["mask"])
    assert err["nocs_coor"] < 2e-2 and err["ivfc_coor"] < 2e-2
    # fp16 operands cannot meet 1e-4 at all -- tests/precision_model.py, DESIGN.md 5c.  Per crop the error of R is small
    # (median 3.5e-3, 90th percentile 9e-3 over the 64 crops); the rot6d -> R normalisation amplifies the fp16 error of the
    # one or two least well-conditioned crops, and THAT maximum is chaotic: numerically equivalent builds (e.g. the fp32-accurate
    # stem on VALU or on MFMA) move it between 2.3e-2 and 6.2e-2.  A regression in any fp16 kernel must not hide behind that
    # crop, so the bound that carries the test is on what the network itself computes -- the rot6d logits BEFORE the
    # normalisation, relative to their scale (measured 4e-3 ... 6e-3) -- plus the distribution of |dR| up to its 99th
    # percentile; the maximum is bounded per crop by that crop's own conditioning (below).
    per_crop = (out["rot"].cpu() - ref["rot"]).abs().reshape(out["rot"].shape[0], -1).max(1).values.sort().values
    r6 = float((rot6d - ref["rot6d"]).abs().max() / ref["rot6d"].abs().max())
    print("fp16 bs64 per-crop |dR|: median %.4f p90 %.4f p99 %.4f max %.4f; rot6d logits rel %.2e" %
          (float(per_crop[32]), float(per_crop[57]), float(per_crop[62]), float(per_crop[-1]), r6))
    assert r6 < 1.5e-2
    assert float(per_crop[32]) < 8e-3 and float(per_crop[57]) < 2e-2 and float(per_crop[62]) < 5e-2
    # the maximum itself: every crop within what ITS logit error and conditioning explain (givepose_amd/rot_cond.py), no fixed ceiling
    from givepose_amd.rot_cond import rot_error_bound
    nB = out["rot"].shape[0]       # (allocentric R: the 6-D -> matrix map the bound is for; the egocentric turn that follows depends on t as well)
    per_u = (dev_out["rot_allo"].float().cpu().reshape(nB, -1) - ref["rot_allo"].reshape(nB, -1)).abs().max(1).values.double()
    # (a crop is excused from the bound only if the REFERENCE's logits alone say it is ill-conditioned for a logit error of the size just
    # asserted; a well-conditioned crop that moved far gets bound 0 and fails; at most 2 of the 64 crops may be excused)
    bound = rot_error_bound(ref["rot6d"], rot6d, max_logit_err=1.5e-2 * float(ref["rot6d"].abs().max()))
    assert int(torch.isinf(bound).sum()) <= 2, int(torch.isinf(bound).sum())
    assert bool((per_u <= bound).all()), float((per_u / bound).max())
    assert err["size"] < 3e-2
    assert err["trans"] < 3e-2 * max(1.0, float(ref["trans"].abs().max()))


@pytest.mark.parametrize("B", [1, 4, 5])
def test_fp16_close_to_reference_golden(golden, net16, B):
    z = golden(f"posenet_e2e_B{B}")
    out = net16(_batch(B, int(z["batch_seed"])), "cuda")
    err = {k: float(np.abs(out[k].cpu().numpy() - z["out_" + k]).max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    print("fp16", B, err)
    assert np.array_equal(out["mask"].cpu().numpy(), z["out_mask"])
    assert err["nocs_coor"] < 2e-2 and err["ivfc_coor"] < 2e-2
    assert err["rot"] < 3e-2 and err["size"] < 3e-2
    assert err["trans"] < 3e-2 * max(1.0, float(np.abs(z["out_trans"]).max()))


@pytest.mark.parametrize("kw", [dict(defer_ln=True), dict(fuse_mlp=False), dict(fuse_mlp_min_batch=1)])
def test_fp16_alternative_block_paths_close_to_reference_golden(golden, kw):
    """The switchable ConvNeXt block paths (LayerNorm folded into fc1's epilogue at C = 512; unfused fc1 / fc2 at
    C = 128 / 256; the fused MLP kernel below its default batch threshold) meet the same fp16 tolerances as the default wiring."""
    z = golden("posenet_e2e_B4")
    net = _model(torch.float16, **kw)
    labels = _launch_labels(net, _batch(4, int(z["batch_seed"])))
    if "defer_ln" in kw:      # 27 stage-2 blocks through the raw depth-wise kernel + LayerNorm folded into fc1's epilogue
        assert sum(n for l, n in labels.items() if "gp_dwconv7_raw_stats" in l) == 27, labels
        assert sum(n for l, n in labels.items() if "N2048 K512 epi6" in l) == 27, labels
    elif "fuse_mlp" in kw:    # no fused MLP launch; stages 0-1 run fc1 / fc2 as GEMMs
        assert not any("convnext_mlp" in l for l in labels), labels
        assert sum(n for l, n in labels.items() if " N512 K128 epi1" in l or " N1024 K256 epi1" in l) == 6, labels
    else:                     # the fused kernel at 4 crops (default: from PoseNetConfig.fuse_mlp_min_batch = 32 crops up)
        assert sum(n for l, n in labels.items() if "convnext_mlp" in l) == 6, labels
        assert not any("convnext_mlp" in l for l in _launch_labels(_model(torch.float16), _batch(4, 3)))
        assert sum(n for l, n in _launch_labels(_model(torch.float16), _batch(32, 3)).items() if "convnext_mlp" in l) == 6
    out = net(_batch(4, int(z["batch_seed"])), "cuda")
    err = {k: float(np.abs(out[k].cpu().numpy() - z["out_" + k]).max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    print("fp16", kw, err)
    assert err["nocs_coor"] < 2e-2 and err["ivfc_coor"] < 2e-2
    assert err["rot"] < 3e-2 and err["size"] < 3e-2
    assert err["trans"] < 3e-2 * max(1.0, float(np.abs(z["out_trans"]).max()))


def test_graph_replay_equals_eager():
    from givepose_amd import PoseNet, PoseNetConfig
    net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True).cuda()
    data = _batch(4, 7)
    a = {k: v.clone() for k, v in net.forward_device(data).items() if k in ("rot", "trans", "size", "ivfc_coor")}   # eager warm-up
    b = {k: v.clone() for k, v in net.forward_device(data).items() if k in a}                                        # capture + replay
    c = {k: v.clone() for k, v in net.forward_device(_batch(4, 8)).items() if k in a}                                 # replay, new inputs
    d = {k: v.clone() for k, v in net.forward_device(data).items() if k in a}
    for k in a:
        assert torch.equal(a[k], b[k]) and torch.equal(a[k], d[k]), k
    assert not torch.equal(a["ivfc_coor"], c["ivfc_coor"])


def test_batches_in_flight_equal_serial():
    """Two slots (own buffers, hipGraph, stream, split-K workspace; shared weights) overlapping on the device give
    bit-identical results to the same batches run one after the other."""
    from givepose_amd import PoseNet, PoseNetConfig
    net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True, inflight=2).cuda()
    with pytest.raises(ValueError):
        net.forward_device(_batch(1, 1), slot=2)
    d0, d1 = _batch(4, 21), _batch(4, 22)
    keys = ("rot", "trans", "size", "nocs_coor", "ivfc_coor")
    ref0 = {k: v.clone() for k, v in net.forward_device(d0).items() if k in keys}
    ref1 = {k: v.clone() for k, v in net.forward_device(d1).items() if k in keys}
    for slot, d in ((0, d0), (1, d1)):           # warm-up + capture of both slots
        for _ in range(2):
            net.forward_device(d, slot=slot)
    torch.cuda.synchronize()
    outs = [None, None]
    for rep in range(10):
        for slot, d in ((0, d0), (1, d1)):
            outs[slot] = net.forward_device(d, slot=slot, wait=False)
        torch.cuda.synchronize()
        for k in keys:
            e0, e1 = float((outs[0][k] - ref0[k]).abs().max()), float((outs[1][k] - ref1[k]).abs().max())
            assert e0 == 0.0 and e1 == 0.0, (rep, k, e0, e1)
    assert net.stream(0) is not net.stream(1)


def test_batches_in_flight_bs64_stress():
    """bs = 64 (the bench shape): three slots overlapping, every buffer that the path writes checked bitwise against the
    serial run of the same slot, 25 repetitions."""
    from givepose_amd import PoseNet, PoseNetConfig, synth
    B, NS = 64, 3
    net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True, inflight=NS).cuda()
    dev = torch.device("cuda")
    d = [{k: torch.from_numpy(v).cuda() for k, v in synth.synth_batch(B, seed=31 + i).items()} for i in range(NS)]
    unwritten = ("h0", "h1", "e_in0", "e_in1", "e_in2")                  # scratch the default wiring never fills completely
    ref = []
    for i in range(NS):
        for _ in range(3):
            net.forward_device(d[i], slot=i)
        torch.cuda.synchronize()
        ref.append({k: v.clone() for k, v in net._plan(B, dev, i)["buf"].items() if k not in unwritten})
    for rep in range(25):
        for i in range(NS):
            net.forward_device(d[i], slot=i, wait=False)
        torch.cuda.synchronize()
        for i in range(NS):
            buf = net._plan(B, dev, i)["buf"]
            bad = [k for k, r in ref[i].items() if not torch.equal(buf[k], r)]
            assert not bad, (rep, i, bad)


def test_use_dcn_off_variant_matches_oracle():
    """BASELINE config 2: MAPEncoder with plain 3x3 s2 convs (use_dcn='')."""
    from givepose_amd import synth
    from givepose_amd.config import PoseNetConfig
    from oracle import posenet_ref as O
    cfg = PoseNetConfig(use_dcn="")
    net = _model(torch.float32, use_dcn="")
    data = _batch(2, 11)
    ref = O.posenet_forward_ref(O.load_params(synth.synth_state_dict(cfg, 0)), data, cfg)
    out = net(data, "cuda")
    for k, tol in (("rot", 1e-4), ("trans", 1e-4), ("size", 1e-4), ("ivfc_coor", 2e-4)):
        assert float((out[k].cpu() - ref[k]).abs().max()) < tol, k


def test_state_dict_contract(golden):
    import json, os
    from givepose_amd import PoseNet
    man = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "state_dict_manifest.json")))
    net = PoseNet()
    sd = net.state_dict()
    ref = man["non_backbone_from_reference"]
    assert [k for k in sd if not k.startswith("backbone.")] == list(ref)
    assert all(list(sd[k].shape) == ref[k] for k in ref)
    net.load_state_dict(sd, strict=True)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float16, 5e-2)])
def test_resnet34_variant_matches_oracle(dtype, tol):
    """a14 / BASELINE configs 1-2: ResNet-34 trunk (network/resnet.py:167-176, pinned by tests/golden/resnet34_trunk.npz
    through the oracle) + the same heads with feature_channel 512 -- the build's own wiring."""
    from givepose_amd import synth
    from givepose_amd.config import PoseNetConfig
    from oracle import posenet_ref as O
    cfg = PoseNetConfig(main_backbone="resnet34")
    net = _model(dtype, main_backbone="resnet34")
    data = _batch(3, 17)
    ref = O.posenet_forward_ref(O.load_params(synth.synth_state_dict(cfg, 0)), data, cfg)
    out = net(data, "cuda")
    err = {k: float((out[k].cpu() - ref[k]).abs().max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    print("resnet34", dtype, err)
    assert all(v < tol for v in err.values()), err


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-4), (torch.float16, 5e-2)])
def test_attention_encoder_variant_matches_oracle(dtype, tol):
    """a13 / BASELINE config 4's in-repo analogue: nocsmap_encoder='att' (MAPTransformerEncoer, 64 tokens, 3 ViT blocks)."""
    from givepose_amd import synth
    from givepose_amd.config import PoseNetConfig
    from oracle import posenet_ref as O
    cfg = PoseNetConfig(nocsmap_encoder="att")
    net = _model(dtype, nocsmap_encoder="att")
    data = _batch(3, 19)
    ref = O.posenet_forward_ref(O.load_params(synth.synth_state_dict(cfg, 0)), data, cfg, return_intermediates=True)
    dev = net.forward_device(data)
    nf = dev["feat_cat"][..., 256:].float().cpu().permute(0, 3, 1, 2)
    print("nocs_feat err", float((nf - ref["nocs_feat"]).abs().max()))
    out = net(data, "cuda")
    err = {k: float((out[k].cpu() - ref[k]).abs().max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    print("att", dtype, err)
    assert all(v < tol for v in err.values()), err


@pytest.mark.parametrize("mode,tol", [("f32", 2e-4), ("split", 2e-4), ("f16", 5e-2)])
def test_attention_encoder_variant_bs32_matches_oracle(mode, tol):
    """BASELINE configs[3] at ITS batch: nocsmap_encoder='att' (the in-repo analogue of the DINOv2 / attention variant, SURVEY.md
    0.2) with 32 crops, hipGraph replay as bench.py runs it, against the oracle -- all three modes."""
    from givepose_amd import PoseNet, synth
    from givepose_amd.config import PoseNetConfig
    from oracle import posenet_ref as O
    cfg = PoseNetConfig(nocsmap_encoder="att")
    kw = {"f32": dict(dtype=torch.float32), "split": dict(dtype=torch.float32, split_gemm=True), "f16": dict(dtype=torch.float16)}[mode]
    net = PoseNet(cfg, seed=0, use_graph=True, **kw).cuda()
    data = _batch(32, 1932)
    torch.set_num_threads(min(16, torch.get_num_threads()))
    ref = O.posenet_forward_ref(O.load_params(synth.synth_state_dict(cfg, 0)), data, cfg)
    for _ in range(3):
        out = net(data, "cuda")
    err = {k: float((out[k].cpu() - ref[k]).abs().max()) for k in ("rot", "trans", "size", "nocs_coor", "ivfc_coor")}
    per = (out["rot"].cpu() - ref["rot"]).abs().reshape(32, -1).max(1).values.sort().values
    print("att bs32", mode, err, "median |dR| %.2e" % float(per[16]))
    assert torch.equal(out["mask"].cpu(), ref["mask"])
    if mode == "f16":      # the worst crop's R is conditioning-bound (see test_fp16_bs64_close_to_oracle): bound the rest tightly
        assert float(per[16]) < 8e-3 and float(per[28]) < 2.5e-2 and err["rot"] < 8e-2
        assert all(err[k] < tol for k in ("trans", "size", "nocs_coor", "ivfc_coor")), err
    else:
        assert err["rot"] < 1e-4 and err["trans"] < 1e-4 and err["size"] < 1e-4 and err["nocs_coor"] < tol and err["ivfc_coor"] < tol, err


def test_bs64_default_wiring_runs_the_tuned_kernels(net16):
    """Dispatch guard: at the bench shape the fp16 path must run the schedules DESIGN.md prices -- stage-2 fc1 with its weights
    in registers (variant 21: two accumulator sets, GELU on packed fp16; round 4: 17), the 3x3 head convs on the LDS-window kernel (13), stages 0-1 on the fused MLP -- and the
    split-operand mode their split forms.  (Round 3 once lost variant 16 to an if / else slip: -5 % end to end, no test noticed.)"""
    from givepose_amd import PoseNet, PoseNetConfig
    lab = _launch_labels(net16, _batch(64, 3))
    n = lambda key: sum(v for l, v in lab.items() if key in l)
    assert n("gemm v21 M16384 N2048 K512 epi1") == 27, lab
    assert n("conv3x3 s1 v13 64x64") == 4 and n("conv3x3 s1 v13 32x32") == 4, lab
    assert n("convnext_mlp C128") == 3 and n("convnext_mlp C256") == 3, lab
    assert n("N512 K2048 epi4") == 27, lab
    sp = _launch_labels(PoseNet(PoseNetConfig(), dtype=torch.float32, seed=0, split_gemm=True).cuda(), _batch(64, 3))
    m = lambda key: sum(v for l, v in sp.items() if key in l)
    assert m("gemm v10 M16384 N2048 K512 epi1 split3") == 27 and m("gemm v7 M16384 N512 K2048 epi4 split3") == 27, sp
    assert m("conv3x3 s1 v13 64x64 Cin256 Cout256 M262144 +gn split3") == 4, sp
    assert m("split_planes") <= 30, sp          # only the small tensors still take a split pass


@pytest.mark.parametrize("B,pins", [
    (4, {"gemm v7 M1024 N2048 K512 epi1": 27, "gemm v18 M1024 N512 K2048 epi4": 27, "conv3x3 s1 v7 64x64": 4, "conv3x3 s1 v18 32x32": 4, "conv3x3 s1 v18 16x16": 4,
         "conv3x3 s1 v18 16x16 Cin256 Cout256 M1024 +gn32": 4, "gemm v23 M4 N2048 K8192 epi3": 1, "gemm v23 M4 N256 K1024 epi3": 2}),
    (8, {"gemm v7 M2048 N2048 K512 epi1": 27, "gemm v18 M2048 N512 K2048 epi4": 27, "conv3x3 s1 v7 64x64": 4, "conv3x3 s1 v7 32x32": 4, "conv3x3 s1 v18 16x16": 4}),
    (16, {"gemm v7 M4096 N2048 K512 epi1": 27, "gemm v7 M4096 N512 K2048 epi4": 27, "conv3x3 s1 v13 64x64": 4, "conv3x3 s1 v7 32x32": 4, "conv3x3 s1 v18 16x16": 4,
          "convnext_mlp C128": 3, "convnext_mlp C256": 3})])       # (round 6: the fused MLP of stages 0 / 1 from 16 crops, the weights-in-registers fc1 from 24)
def test_small_batch_wiring_is_pinned(net16, B, pins):
    """Dispatch guard for the batches between the latency path and the bench shape (round-4 advice): the automatic choice between the latency
    kernel (variant 18, cost model fitted on 1-8 crops, capped at 16 384 rows) and the tile kernels moves a whole forward by tens of percent
    and no numerics test notices.  The launches that carry a forward at 4 / 8 / 16 crops, as measured when the model was fitted
    (profiles/r04_small_m_tiles.txt; the labels of round 5: scripts/dump_labels.py)."""
    lab = _launch_labels(net16, _batch(B, 3))
    for key, n in pins.items():
        assert sum(v for l, v in lab.items() if key in l) == n, (key, lab)
    assert not any(" v18 " in l and int(re.search(r" M(\d+)", l).group(1)) > 16384 for l in lab), lab


def test_grouped_launches_in_flight_bs128_stress():
    """The bench default: two launch sequences of 2 x 64 crops in flight; every buffer the path writes checked bitwise against
    the serial run of the same slot, 15 repetitions."""
    from givepose_amd import PoseNet, PoseNetConfig, synth
    B, NS = 128, 2
    net = PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True, inflight=NS, dcn_couple=64).cuda()
    dev = torch.device("cuda")
    d = [{k: torch.from_numpy(v).cuda() for k, v in synth.synth_batch(B, seed=51 + i).items()} for i in range(NS)]
    unwritten = ("h0", "h1", "e_in0", "e_in1", "e_in2")                  # scratch the default wiring never fills completely
    ref = []
    for i in range(NS):
        for _ in range(3):
            net.forward_device(d[i], slot=i)
        torch.cuda.synchronize()
        ref.append({k: v.clone() for k, v in net._plan(B, dev, i)["buf"].items() if k not in unwritten})
    for rep in range(15):
        for i in range(NS):
            net.forward_device(d[i], slot=i, wait=False)
        torch.cuda.synchronize()
        for i in range(NS):
            buf = net._plan(B, dev, i)["buf"]
            bad = [k for k, r in ref[i].items() if not torch.equal(buf[k], r)]
            assert not bad, (rep, i, bad)


def test_fp16_with_fp32_residual_stream(oracle64):
    """PoseNetConfig(res_fp32=True): the residual stream of ConvNeXt stage 2 accumulated in fp32 (fp32 residual / output of the
    downsample conv and of every fc2, an fp16 copy for the depth-wise conv, fp32 rows into the next downsample LayerNorm).  Same
    launches otherwise; the trunk feature must come out closer to the oracle's than the plain fp16 mode's, the poses at least as close
    in the median; the model of tests/precision_model.py says 1.7 x on R."""
    from givepose_amd import PoseNet, PoseNetConfig
    data, ref = oracle64("dcnv3")
    errs = {}
    for name, cfg in (("fp16", PoseNetConfig()), ("res32", PoseNetConfig(res_fp32=True))):
        net = PoseNet(cfg, dtype=torch.float16, seed=0, use_graph=True).cuda()
        for _ in range(3):
            out = net(data, "cuda")
        dev_out = net.forward_device(data)
        feat = dev_out["feat"].float().cpu().permute(0, 3, 1, 2)
        per = (out["rot"].cpu() - ref["rot"]).abs().reshape(64, -1).max(1).values.sort().values
        errs[name] = {"feat": float((feat - ref["feat"]).abs().max()), "feat_mean": float((feat - ref["feat"]).abs().mean()),
                      "rot_median": float(per[32]), "rot_p90": float(per[57]), "rot_max": float(per[-1]),
                      "size": float((out["size"].cpu() - ref["size"]).abs().max())}
        if name == "res32":       # (labels need eager launches: a hipGraph replay carries no per-launch hooks)
            labels = _launch_labels(PoseNet(cfg, dtype=torch.float16, seed=0).cuda(), _batch(4, 3))
            assert sum(v for l, v in labels.items() if "N512 K2048 epi4" in l) == 27, labels
    print("fp32 residual stream:", errs)
    assert errs["res32"]["feat_mean"] < 0.8 * errs["fp16"]["feat_mean"]
    assert errs["res32"]["rot_median"] < 1.1 * errs["fp16"]["rot_median"] and errs["res32"]["rot_p90"] < 2e-2
