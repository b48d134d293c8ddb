"""GPU: the per-frame body of evaluation/evaluate.py:100-126 strung together from the drop-in pieces
(givepose_amd.pipeline.FramePipeline: gp_crop_rois -> Scale_net -> PoseNet -> gp_pred_rt) against the same chain of
oracles (oracle/preprocess_ref.py -> scale_net_ref -> posenet_ref -> pred_rt_ref), plus the checkpoint key remapper."""
import numpy as np
import pytest
import torch

T = torch.from_numpy


@pytest.mark.gpu
def test_frame_pipeline_matches_oracle_chain():
    from givepose_amd import PoseNet, PoseNetConfig, Scale_net, synth
    from givepose_amd.pipeline import FramePipeline
    from oracle import posenet_ref as O, preprocess_ref as PR, scale_net_ref as S
    rng = np.random.default_rng(12)
    H, W, n = 480, 640, 5
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    masks = (rng.random((n, H, W)) > 0.5).astype(np.uint8)
    y1, x1 = rng.integers(0, 200, n), rng.integers(0, 300, n)
    boxes = np.stack([y1, x1, y1 + rng.integers(60, 260, n), x1 + rng.integers(60, 320, n)], 1)
    cats = rng.integers(0, 6, n)
    mean_shapes = synth.MEAN_SIZES[cats]
    full = rng.standard_normal((3, 256, 256)).astype(np.float32)
    cfg = PoseNetConfig()
    net = PoseNet(cfg, dtype=torch.float32, seed=0).cuda()
    sn = Scale_net(feat_dim=24, seed=0).cuda()
    rt, size, out = FramePipeline(net, sn)(frame, masks, boxes, cats, synth.REAL_INTRINSICS, mean_shapes, full)
    # ---- the same chain on the CPU oracles
    d = PR.crop_batch_ref(frame, np.moveaxis(masks, 0, 2), boxes)
    data = {k: T(v) for k, v in d.items()}
    data["cam_K"] = T(np.broadcast_to(synth.REAL_INTRINSICS, (n, 3, 3)).copy())
    data["mean_size"] = T(mean_shapes.copy())
    data["full_img"] = T(np.broadcast_to(full, (n, 3, 256, 256)).copy())
    data["one_hot"] = T(np.eye(6, dtype=np.float32)[cats])
    with torch.no_grad():
        scale = S.scale_net_forward_ref({k: T(v) for k, v in synth.synth_scale_net_state_dict(24, 0).items()}, data)
        ref = O.posenet_forward_ref(O.load_params(synth.synth_state_dict(cfg, 0)), data, cfg)
    ref_rt, ref_size = PR.pred_rt_ref(ref["rot"].numpy(), ref["trans"].numpy(), ref["size"].numpy(), scale.numpy())
    assert torch.equal(out["mask"].cpu(), ref["mask"])
    assert np.abs(rt.cpu().numpy() - ref_rt).max() < 2e-4 * max(1.0, np.abs(ref_rt).max())
    assert np.abs(size.cpu().numpy() - ref_size).max() < 1e-4


@pytest.mark.gpu
def test_run_frames_equals_frame_by_frame():
    """FramePipeline.run_frames (the detections of several frames in ONE launch sequence, per-frame DCNv3 coupling through
    PoseNet.forward_device(groups=...)) against the one-frame pipeline called once per frame (itself checked against the oracle chain above)."""
    from givepose_amd import PoseNet, PoseNetConfig, Scale_net, synth
    from givepose_amd.pipeline import FramePipeline
    rng = np.random.default_rng(13)
    H, W, sizes = 480, 640, (3, 1, 6, 2)
    F = len(sizes)
    frames = rng.integers(0, 256, (F, H, W, 3), dtype=np.uint8)
    masks = [(rng.random((n, H, W)) > 0.5).astype(np.uint8) for n in sizes]
    boxes = []
    for n in sizes:
        y1, x1 = rng.integers(0, 200, n), rng.integers(0, 300, n)
        boxes.append(np.stack([y1, x1, y1 + rng.integers(60, 260, n), x1 + rng.integers(60, 320, n)], 1))
    cats = [rng.integers(0, 6, n) for n in sizes]
    shapes = [synth.MEAN_SIZES[c] for c in cats]
    full = rng.standard_normal((F, 3, 256, 256)).astype(np.float32)
    net = PoseNet(PoseNetConfig(), dtype=torch.float32, seed=0).cuda()
    pipe = FramePipeline(net, Scale_net(feat_dim=24, seed=0).cuda())
    rt, size, out, got_sizes = pipe.run_frames(frames, masks, boxes, cats, synth.REAL_INTRINSICS, shapes, full)
    rt, size, mask = rt.clone(), size.clone(), out["mask"].clone()
    assert list(got_sizes) == list(sizes) and rt.shape[0] == sum(sizes)
    i = 0
    for f, n in enumerate(sizes):
        rt1, size1, out1 = pipe(frames[f], masks[f], boxes[f], cats[f], synth.REAL_INTRINSICS, shapes[f], full[f])
        assert torch.equal(mask[i:i + n], out1["mask"])
        assert float((rt[i:i + n] - rt1).abs().max()) < 1e-4 * max(1.0, float(rt1.abs().max())), f
        assert float((size[i:i + n] - size1).abs().max()) < 5e-5, f
        i += n


def test_checkpoint_key_remap_cpu():
    """timm un-flattened names, HuggingFace names and a DataParallel prefix land on the registered names; junk is refused."""
    from givepose_amd import PoseNet
    from givepose_amd.checkpoint import load_checkpoint, remap_keys
    net = PoseNet(seed=0)
    sd = {k: v.clone() for k, v in net.state_dict().items()}      # state_dict() aliases the live tensors
    alt = {}
    for k, v in sd.items():
        if k.startswith("backbone.stem_"):
            k2 = "module." + k.replace("backbone.stem_0", "backbone.stem.0").replace("backbone.stem_1", "backbone.stem.1")
        elif k.startswith("backbone.stages_"):
            k2 = k.replace("backbone.stages_", "backbone.stages.", 1)
        else:
            k2 = "module." + k
        alt[k2] = v.clone() + (1 if v.is_floating_point() else 0)
    hf = {"backbone.encoder.stages.2.layers.5.pwconv1.weight": 0, "backbone.embeddings.patch_embeddings.weight": 0}
    assert set(remap_keys(hf)[0]) == {"backbone.stages_2.blocks.5.mlp.fc1.weight", "backbone.stem_0.weight"}
    rep = load_checkpoint(net, alt, verbose=False)
    assert not rep["unknown"] and not rep["missing"] and len(rep["renamed"]) == len(alt)
    assert all(torch.equal(net.state_dict()[k], sd[k] + (1 if sd[k].is_floating_point() else 0)) for k in list(sd)[:40])
    part = {k: v for k, v in list(sd.items())[:10]}                      # partial checkpoint: evaluate.py:53-55 semantics
    assert len(load_checkpoint(net, part, verbose=False)["missing"]) == len(sd) - 10
    with pytest.raises(KeyError):
        load_checkpoint(net, {"backbone.not_a_layer.weight": torch.zeros(1)}, verbose=False)
    with pytest.raises(ValueError):
        load_checkpoint(net, {"feat_reducer.bias": torch.zeros(7)}, verbose=False)
