"""givepose_amd -- MI355X-native (gfx950) inference path for GIVEPose's PoseNet.

Public surface mirrors the reference for this path:
  PoseNet            network/PoseNet.py:134-231 (forward(data, device, do_loss=False, pred_scale=None) -> dict)
  dcnv3_forward      the pybind op DCNv3.dcnv3_forward (network/ops_dcnv3/src/dcnv3.h:20-38)
  PoseNetConfig      the absl FLAGS the path reads (config/config.py)
"""
from .config import PoseNetConfig  # noqa: F401
from .posenet import PoseNet  # noqa: F401


def dcnv3_forward(*args, **kwargs):
    from .ops import dcnv3_forward as f
    return f(*args, **kwargs)
