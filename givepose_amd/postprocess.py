"""Evaluation post-processing on the device: what evaluation/evaluate.py:116-125 does with the
model's output dict on the CPU -- pred_RT = [[R | t] * pred_scale; 0 0 0 1] and the L2-normalised size."""
import torch

from . import _lib


def pred_rt(out, pred_scale=None):
    """out: PoseNet.forward_device(...) dict (device-resident rot (B,3,3), trans (B,3), size (B,3)); pred_scale (B,)
    or None -> (pred_RT (B,4,4), pred_size (B,3)) fp32 on the same device, asynchronous on the current stream."""
    R, t, s = out["rot"], out["trans"], out["size"]
    if R.device.type != "cuda":
        raise RuntimeError("pred_rt needs the device-resident output of PoseNet.forward_device (no CPU fallback)")
    B, dev = R.shape[0], R.device
    R = R.reshape(B, 9).to(torch.float32).contiguous()
    t = t.to(dev, torch.float32).contiguous()
    s = s.to(dev, torch.float32).contiguous()
    sc = None if pred_scale is None else torch.as_tensor(pred_scale).to(dev, torch.float32).reshape(B).contiguous()
    rt = torch.empty(B, 4, 4, device=dev)
    ps = torch.empty(B, 3, device=dev)
    L = _lib.load()
    _lib.check(L.gp_pred_rt(R.data_ptr(), t.data_ptr(), s.data_ptr(), 0 if sc is None else sc.data_ptr(), rt.data_ptr(),
                            ps.data_ptr(), B, torch.cuda.current_stream(dev).cuda_stream), "gp_pred_rt")
    return rt, ps
