"""ORACLE -- test infrastructure, not product code.

CPU restatement (plain PyTorch fp32) of the reference ``Scale_net.forward`` (network/scale_net.py:45-65): two
torchvision ``mobilenet_v3_small`` feature extractors (+ avgpool + flatten), dropout (eval: identity), line1 -> ReLU ->
cat(one_hot) -> line2 -> ReLU -> cat(one_hot) -> cat(roi_wh / 100) -> line3, + ||mean_size||.

torchvision 0.15.2 is a third-party dependency that is absent here (GIVEPose_env.yml pins it; not vendored): the
MobileNetV3-small arithmetic is restated from the paper / torchvision's published block settings (Conv-BN(eps 1e-3)-act,
inverted residual with optional SqueezeExcitation(ReLU, Hardsigmoid), Hardswish, residual when stride 1 and in == out).
**Parity unpinned** against torchvision itself.  The Scale_net wiring around it IS pinned: tests/golden/scale_net.npz is
the reference class run on CPU with scripts/ref_shim.py's stand-in for the torchvision model (scripts/gen_golden_scale_net.py).
"""
import torch
import torch.nn.functional as F

from givepose_amd.synth import MBV3S


def _bn(P, x, p):
    return F.batch_norm(x, P[p + ".running_mean"], P[p + ".running_var"], P[p + ".weight"], P[p + ".bias"], False, 0.0, 1e-3)


def _act(x, a):
    return F.relu(x) if a == "RE" else F.hardswish(x)


def mobilenet_v3_small_features_ref(P, x, f):
    """features (13 entries) -> AdaptiveAvgPool2d(1) -> Flatten: (B,3,H,W) -> (B,576)."""
    x = F.hardswish(_bn(P, F.conv2d(x, P[f + ".0.0.weight"], None, stride=2, padding=1), f + ".0.1"))
    for i, (cin, k, exp, cout, se, a, s) in enumerate(MBV3S, 1):
        y, j = x, 0
        if exp != cin:
            y = _act(_bn(P, F.conv2d(y, P[f"{f}.{i}.block.{j}.0.weight"]), f"{f}.{i}.block.{j}.1"), a)
            j += 1
        y = _act(_bn(P, F.conv2d(y, P[f"{f}.{i}.block.{j}.0.weight"], None, stride=s, padding=k // 2, groups=exp), f"{f}.{i}.block.{j}.1"), a)
        j += 1
        if se:
            q = y.mean((2, 3), keepdim=True)
            q = F.relu(F.conv2d(q, P[f"{f}.{i}.block.{j}.fc1.weight"], P[f"{f}.{i}.block.{j}.fc1.bias"]))
            q = F.hardsigmoid(F.conv2d(q, P[f"{f}.{i}.block.{j}.fc2.weight"], P[f"{f}.{i}.block.{j}.fc2.bias"]))
            y = y * q
            j += 1
        y = _bn(P, F.conv2d(y, P[f"{f}.{i}.block.{j}.0.weight"]), f"{f}.{i}.block.{j}.1")
        x = x + y if (s == 1 and cin == cout) else y
    x = F.hardswish(_bn(P, F.conv2d(x, P[f + ".12.0.weight"]), f + ".12.1"))
    return x.mean((2, 3))


def scale_net_forward_ref(P, data, use_hw=True):
    """network/scale_net.py:45-65.  Returns scale (B,)."""
    fr = mobilenet_v3_small_features_ref(P, data["roi_img"].float(), "feat_encoder_bbox.0")
    ff = mobilenet_v3_small_features_ref(P, data["full_img"].float(), "feat_encoder_full.0")
    one_hot = data["one_hot"].float()
    x = F.relu(F.linear(torch.cat([fr, ff], 1), P["line1.weight"], P["line1.bias"]))
    x = F.relu(F.linear(torch.cat([x, one_hot], 1), P["line2.weight"], P["line2.bias"]))
    x = torch.cat([x, one_hot], 1)
    if use_hw:
        x = torch.cat([x, data["roi_wh"].float() / 100], 1)
    resi = F.linear(x, P["line3.weight"], P["line3.bias"]).squeeze(-1)
    return resi + data["mean_size"].float().norm(dim=1)
