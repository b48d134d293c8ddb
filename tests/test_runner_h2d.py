"""GPU: the pipelined host -> device path of givepose_amd.runner.ShardRunner (copy stream -> per-slot staging -> static inputs,
slots in flight) must change nothing: poses bit for bit those of the same batches resident in HBM; the uint8-frames form equals
RoiCropper + forward on the same detections."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _net(nf):
    from givepose_amd import PoseNet, PoseNetConfig
    return PoseNet(PoseNetConfig(), dtype=torch.float16, seed=0, use_graph=True, inflight=nf).cuda()


def test_h2d_crops_equal_resident():
    from givepose_amd import synth
    from givepose_amd.runner import ShardRunner
    B, NF = 8, 2
    dev = torch.device("cuda", torch.cuda.current_device())
    net = _net(NF)
    batches = [synth.synth_batch(B, seed=40 + i) for i in range(NF)]
    res = ShardRunner(net, B, dev, 1, inflight=NF)
    for i in range(NF):
        res.load(i, batches[i])
    for _ in range(3 * NF):
        res.step()
    torch.cuda.synchronize()
    want = [res.result(i).clone() for i in range(NF)]
    run = ShardRunner(net, B, dev, 1, inflight=NF, h2d="crops")
    for i in range(NF):
        run.load(i, batches[i])
        for k in run.statics[i]:                      # poison the resident inputs: every step must really bring its own
            run.statics[i][k].fill_(float("nan"))
    for _ in range(4 * NF + 1):
        run.step()
    torch.cuda.synchronize()
    for i in range(NF):
        assert torch.equal(run.result(i), want[i]), i
    assert run.host_bytes == sum(v.size * 4 for v in batches[0].values())


def test_h2d_frames_equal_cropper_then_forward():
    from givepose_amd import synth
    from givepose_amd.preprocess import RoiCropper
    from givepose_amd.runner import ShardRunner
    from givepose_amd import dist as gd
    B, NF = 8, 2
    dev = torch.device("cuda", torch.cuda.current_device())
    net = _net(NF)
    rng = np.random.default_rng(5)
    sets = []
    for i in range(NF):
        frames = rng.integers(0, 256, (2, 480, 640, 3), dtype=np.uint8)
        masks = (rng.random((B, 480, 640)) > 0.5).astype(np.uint8)
        y1, x1 = rng.integers(0, 200, B), rng.integers(0, 300, B)
        boxes = np.stack([y1, x1, y1 + rng.integers(60, 260, B), x1 + rng.integers(60, 320, B)], axis=1)
        scal = {k: v for k, v in synth.synth_batch(B, seed=60 + i).items() if k in ("cam_K", "mean_size")}
        sets.append((frames, masks, [j // 4 for j in range(B)], list(range(B)), boxes, scal))
    run = ShardRunner(net, B, dev, 1, inflight=NF, h2d="frames")
    for i, s in enumerate(sets):
        run.load_frames(i, *s)
    for _ in range(4 * NF + 1):
        run.step()
    torch.cuda.synchronize()
    got = [run.result(i).clone() for i in range(NF)]
    ref_net = net                      # the same schedules (a net built for one batch in flight picks other tiles: last-bit differences)
    cropper = RoiCropper(480, 640, dev)
    for i, (frames, masks, fi, mi, boxes, scal) in enumerate(sets):
        st = ref_net.static_inputs(B, dev)
        for k in st:
            st[k].zero_()
        for k, v in scal.items():
            st[k].copy_(torch.as_tensor(v).reshape(st[k].shape))
        cropper(torch.from_numpy(frames).to(dev), torch.from_numpy(masks).to(dev), fi, mi, boxes, out=st)
        o = ref_net.forward_device(st, dev)
        torch.cuda.synchronize()
        assert torch.equal(gd.pack_poses(o["rot"], o["trans"], o["size"]), got[i]), i
