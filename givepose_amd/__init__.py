"""givepose_amd -- MI355X-native (gfx950) inference path for GIVEPose's PoseNet.

Public surface mirrors the reference for this path:
  PoseNet            network/PoseNet.py:134-231 (forward(data, device, do_loss=False, pred_scale=None) -> dict)
  dcnv3_forward      the pybind op DCNv3.dcnv3_forward (network/ops_dcnv3/src/dcnv3.h:20-38)
  dcnv3_backward     the pybind op DCNv3.dcnv3_backward (dcnv3.h:40-59);  DCNv3Function: functions/dcnv3_func.py:25-98
  Scale_net          network/scale_net.py:22-65 (forward(data, device, mode) -> scale (B,)), run before PoseNet by evaluate.py
  PoseNetConfig      the absl FLAGS the path reads (config/config.py)
"""
from .config import PoseNetConfig  # noqa: F401
from .posenet import PoseNet  # noqa: F401


def dcnv3_forward(*args, **kwargs):
    from .ops import dcnv3_forward as f
    return f(*args, **kwargs)


def dcnv3_backward(*args, **kwargs):
    """The pybind op DCNv3.dcnv3_backward (network/ops_dcnv3/src/dcnv3.h:40-59)."""
    from .ops import dcnv3_backward as f
    return f(*args, **kwargs)


def __getattr__(name):
    if name == "DCNv3Function":          # functions/dcnv3_func.py:25-98
        from .dcnv3_function import DCNv3Function
        return DCNv3Function
    if name == "Scale_net":              # network/scale_net.py:22-65
        from .scale_net import Scale_net
        return Scale_net
    raise AttributeError(name)
