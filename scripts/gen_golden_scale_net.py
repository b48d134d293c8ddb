#!/usr/bin/env python3
"""tests/golden/scale_net.npz: the reference's own ``Scale_net`` class (network/scale_net.py:22-65) run on CPU with the
seeded synthetic weights, torchvision's mobilenet_v3_small replaced by scripts/ref_shim.py's stand-in (torchvision is
not installed: the fixture pins the Scale_net wiring -- encoders, concatenations, roi_wh / 100, + ||mean_size|| -- and
the state_dict key layout of that wiring, not torchvision's arithmetic).  Build container only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ref_shim  # noqa: E402

ref_shim.install()
import types  # noqa: E402
import torch  # noqa: E402

# network/scale_net.py imports its training-only dependencies at module level
sys.modules.setdefault("torch.utils.tensorboard", types.SimpleNamespace(SummaryWriter=object))
sys.modules.setdefault("datasets.load_data_nocs", types.SimpleNamespace(NocsDataset=object))
from network.scale_net import Scale_net  # noqa: E402

from givepose_amd import synth  # noqa: E402
from oracle import scale_net_ref as S  # noqa: E402

torch.autograd.set_detect_anomaly(False)
B, FEAT = 3, 24
net = Scale_net(feat_dim=FEAT, use_hw=True, backbone="mobilenetv3s", pretrained=False).eval()
sd = net.state_dict()
man = synth.scale_net_manifest(FEAT)
assert list(sd.keys()) == list(man.keys()), [k for k in sd if k not in man][:5] + [k for k in man if k not in sd][:5]
assert all(tuple(sd[k].shape) == tuple(man[k]) for k in sd)
syn = synth.synth_scale_net_state_dict(FEAT, 0)
net.load_state_dict({k: torch.from_numpy(v) for k, v in syn.items()}, strict=True)
data = {k: torch.from_numpy(v) for k, v in synth.synth_scale_batch(B, seed=77).items()}
with torch.no_grad():
    exp = net(data, "cpu", "test")
P = {k: torch.from_numpy(v) for k, v in syn.items()}
with torch.no_grad():
    got = S.scale_net_forward_ref(P, data)
print("reference class vs oracle:", float((got - exp).abs().max()), exp)
assert float((got - exp).abs().max()) < 1e-5
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "scale_net.npz"), batch_seed=77, B=B, feat_dim=FEAT, expected=exp.numpy(),
                    keys=np.array(list(sd.keys())))
print("wrote scale_net.npz")
