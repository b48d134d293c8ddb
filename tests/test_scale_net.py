"""Scale_net: oracle vs the reference class on CPU, HIP vs golden / oracle on the GPU."""
import numpy as np
import pytest
import torch

T = torch.from_numpy


def test_oracle_matches_reference_class_golden(golden):
    """tests/golden/scale_net.npz = network/scale_net.py's own class (torchvision model replaced by the shim's stand-in)."""
    from givepose_amd import synth
    from oracle import scale_net_ref as S
    z = golden("scale_net")
    feat = int(z["feat_dim"])
    assert list(synth.scale_net_manifest(feat).keys()) == [str(k) for k in z["keys"]]        # the reference's state_dict order
    P = {k: T(v) for k, v in synth.synth_scale_net_state_dict(feat, 0).items()}
    data = {k: T(v) for k, v in synth.synth_scale_batch(int(z["B"]), seed=int(z["batch_seed"])).items()}
    with torch.no_grad():
        got = S.scale_net_forward_ref(P, data).numpy()
    assert np.abs(got - z["expected"]).max() < 1e-5


@pytest.mark.gpu
def test_hip_matches_golden_and_state_dict_contract(golden):
    from givepose_amd import Scale_net, synth
    z = golden("scale_net")
    net = Scale_net(feat_dim=int(z["feat_dim"]), use_hw=True, backbone="mobilenetv3s", seed=0).cuda()
    assert list(net.state_dict().keys()) == [str(k) for k in z["keys"]]
    net.load_state_dict(net.state_dict(), strict=True)
    data = {k: T(v) for k, v in synth.synth_scale_batch(int(z["B"]), seed=int(z["batch_seed"])).items()}
    got = net(data, "cuda", "test")
    assert tuple(got.shape) == (int(z["B"]),) and got.device.type == "cuda"
    assert np.abs(got.cpu().numpy() - z["expected"]).max() < 1e-4
    with pytest.raises(RuntimeError):
        net(data, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("B,size", [(1, 256), (5, 192)])
def test_hip_matches_oracle_other_shapes(B, size):
    """Other batch sizes / image sizes than the fixture (full_img is the un-resized frame when resize_full is off:
    load_data_eval.py:336) against the oracle on this host, incl. the hand-off to PoseNet's pred_scale argument."""
    from givepose_amd import Scale_net, synth
    from oracle import scale_net_ref as S
    net = Scale_net(feat_dim=24, seed=1).cuda()
    data = {k: T(v) for k, v in synth.synth_scale_batch(B, seed=5, img_size=size).items()}
    P = {k: T(v) for k, v in synth.synth_scale_net_state_dict(24, 1).items()}
    with torch.no_grad():
        ref = S.scale_net_forward_ref(P, data)
    got = net(data, "cuda")
    assert float((got.cpu() - ref).abs().max()) < 1e-4
