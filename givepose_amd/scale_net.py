"""Scale_net on HIP kernels -- drop-in for the reference ``network.scale_net.Scale_net`` (network/scale_net.py:22-65;
built and called at evaluation/evaluate.py:61, 111-113): ``Scale_net(feat_dim, use_hw, ...)``, ``forward(data, device,
mode='') -> scale (B,)`` from ``data['roi_img' | 'full_img' | 'one_hot' | 'roi_wh' | 'mean_size']``, and the reference's
state_dict names (two ``nn.Sequential(mobilenet_v3_small.features, avgpool, Flatten)`` encoders + line1..3;
tests/golden/scale_net.npz pins the key list against the reference class).  torchvision's MobileNetV3-small is
third-party arithmetic restated from the published architecture (givepose_amd.synth.MBV3S): parity against torchvision
itself is unpinned.  All arithmetic runs in libgivepose_hip.so (csrc/scalenet.hip, fp32); there is no fallback.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib, synth
from .posenet import _BUFFER_SUFFIXES, _register

_ACT = {"RE": 1, "HS": 2}


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class Scale_net(nn.Module):
    def __init__(self, feat_dim=8, use_hw=True, backbone="mobilenetv3s", pretrained=True, cats_num=6, seed=None):
        super().__init__()
        if backbone != "mobilenetv3s":        # the reference builds mobilenet_v3_small whatever `backbone` says (scale_net.py:25-26)
            pass
        self.feat_dim, self.use_hw, self.cats_num = feat_dim, use_hw, cats_num
        for name, shape in synth.scale_net_manifest(feat_dim, cats_num, use_hw).items():
            is_buf = name.endswith(_BUFFER_SUFFIXES)
            if seed is None:
                val = torch.zeros(shape, dtype=torch.int64 if name.endswith("num_batches_tracked") else torch.float32)
                if name.endswith("running_var"):
                    val += 1
            else:
                val = torch.from_numpy(synth.synth_tensor("scale_net." + name, shape, seed))
            _register(self, name, val if is_buf else nn.Parameter(val, requires_grad=False), not is_buf)
        self._packed = None
        self.eval()

    def load_state_dict(self, state_dict, strict=True):
        r = super().load_state_dict(state_dict, strict=strict)
        self._packed = None
        return r

    def _apply(self, fn, *a, **k):
        r = super()._apply(fn, *a, **k)
        self._packed = None
        return r

    @torch.no_grad()
    def _pack(self, device):
        """Fold eval BatchNorm (eps 1e-3) into the convolutions and lay the weights out for csrc/scalenet.hip."""
        sd = {k: v.detach().to(device="cpu", dtype=torch.float32) for k, v in self.state_dict().items() if v.is_floating_point()}   # folds on the host
        dev = lambda t: t.contiguous().to(device)

        def fold(conv, bn):
            sc = sd[bn + ".weight"] / torch.sqrt(sd[bn + ".running_var"] + 1e-3)
            return sd[conv + ".weight"] * sc[:, None, None, None], sd[bn + ".bias"] - sd[bn + ".running_mean"] * sc

        W = {}
        for enc in ("feat_encoder_bbox", "feat_encoder_full"):
            f, E = enc + ".0", {}
            w, b = fold(f + ".0.0", f + ".0.1")
            E["stem"] = (dev(w.reshape(16, 27).t()), dev(b))
            blocks = []
            for i, (cin, k, exp, cout, se, act, stride) in enumerate(synth.MBV3S, 1):
                j, blk = 0, {"cfg": (cin, k, exp, cout, se, _ACT[act], stride)}
                if exp != cin:
                    w, b = fold(f"{f}.{i}.block.{j}.0", f"{f}.{i}.block.{j}.1")
                    blk["expand"] = (dev(w.reshape(exp, cin)), dev(b))
                    j += 1
                w, b = fold(f"{f}.{i}.block.{j}.0", f"{f}.{i}.block.{j}.1")
                blk["dw"] = (dev(w.reshape(exp, k * k).t()), dev(b))
                j += 1
                if se:
                    q = f"{f}.{i}.block.{j}."
                    sq = sd[q + "fc1.weight"].shape[0]
                    blk["se"] = (dev(sd[q + "fc1.weight"].reshape(sq, exp)), dev(sd[q + "fc1.bias"]),
                                 dev(sd[q + "fc2.weight"].reshape(exp, sq)), dev(sd[q + "fc2.bias"]), sq)
                    j += 1
                w, b = fold(f"{f}.{i}.block.{j}.0", f"{f}.{i}.block.{j}.1")
                blk["project"] = (dev(w.reshape(cout, exp)), dev(b))
                blocks.append(blk)
            E["blocks"] = blocks
            w, b = fold(f + ".12.0", f + ".12.1")
            E["last"] = (dev(w.reshape(synth.MBV3S_LAST, 96)), dev(b))
            W[enc] = E
        for n in ("line1", "line2", "line3"):
            W[n] = (dev(sd[n + ".weight"]), dev(sd[n + ".bias"]))
        self._packed = W
        return W

    def _encoder(self, E, img, lib, st):
        """mobilenet_v3_small.features -> avgpool -> flatten: (B,3,H,W) fp32 -> (B,576)."""
        B, _, H, W_ = img.shape
        dev = img.device
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        h, w = H // 2, W_ // 2
        x = e(B, h, w, 16)
        _lib.check(lib.gp_sn_stem(_p(img), _p(E["stem"][0]), _p(E["stem"][1]), _p(x), B, H, W_, st), "gp_sn_stem")
        for blk in E["blocks"]:
            cin, k, exp, cout, se, act, stride = blk["cfg"]
            y = x
            if "expand" in blk:
                y = e(B, h, w, exp)
                _lib.check(lib.gp_sn_pointwise(_p(x), _p(blk["expand"][0]), _p(blk["expand"][1]), None, None, _p(y), B * h * w, exp, cin, h * w, act, st),
                           "gp_sn_pointwise")
            ho, wo = (h + stride - 1) // stride, (w + stride - 1) // stride
            d = e(B, ho, wo, exp)
            _lib.check(lib.gp_sn_depthwise(_p(y), _p(blk["dw"][0]), _p(blk["dw"][1]), _p(d), B, h, w, exp, k, stride, act, st), "gp_sn_depthwise")
            gate = None
            if se:
                pooled, gate = e(B, exp), e(B, exp)
                _lib.check(lib.gp_sn_avgpool(_p(d), _p(pooled), B, ho * wo, exp, st), "gp_sn_avgpool")
                w1, b1, w2, b2, sq = blk["se"]
                _lib.check(lib.gp_sn_se(_p(pooled), _p(w1), _p(b1), _p(w2), _p(b2), _p(gate), B, exp, sq, st), "gp_sn_se")
            out = e(B, ho, wo, cout)
            res = x if (stride == 1 and cin == cout) else None
            _lib.check(lib.gp_sn_pointwise(_p(d), _p(blk["project"][0]), _p(blk["project"][1]), _p(gate), _p(res), _p(out), B * ho * wo, cout, exp,
                                           ho * wo, 0, st), "gp_sn_pointwise")
            x, h, w = out, ho, wo
        last = e(B, h, w, synth.MBV3S_LAST)
        _lib.check(lib.gp_sn_pointwise(_p(x), _p(E["last"][0]), _p(E["last"][1]), None, None, _p(last), B * h * w, synth.MBV3S_LAST, 96, h * w, 2, st),
                   "gp_sn_pointwise")
        feat = e(B, synth.MBV3S_LAST)
        _lib.check(lib.gp_sn_avgpool(_p(last), _p(feat), B, h * w, synth.MBV3S_LAST, st), "gp_sn_avgpool")
        return feat

    @torch.no_grad()
    def forward(self, data, device="cuda", mode=""):
        """Reference signature (network/scale_net.py:45).  Dropout is the identity in eval (scale_net.py:50,52)."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("givepose_amd.Scale_net runs on the HIP device only (no CPU path)")
        lib = _lib.load()
        if self._packed is None:
            self._pack(device)
        W = self._packed
        st = ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        g = lambda k: data[k].to(device=device, dtype=torch.float32).contiguous()
        roi, full, one_hot, wh, ms = g("roi_img"), g("full_img"), g("one_hot"), g("roi_wh"), g("mean_size")
        B = roi.shape[0]
        f_roi = self._encoder(W["feat_encoder_bbox"], roi, lib, st)
        f_full = self._encoder(W["feat_encoder_full"], full, lib, st)
        scale = torch.empty(B, dtype=torch.float32, device=device)
        _lib.check(lib.gp_sn_head(_p(f_roi), _p(f_full), _p(one_hot), _p(wh), _p(ms), _p(W["line1"][0]), _p(W["line1"][1]), _p(W["line2"][0]),
                                  _p(W["line2"][1]), _p(W["line3"][0]), _p(W["line3"][1]), _p(scale), B, synth.MBV3S_LAST, self.feat_dim,
                                  self.cats_num, 1 if self.use_hw else 0, st), "gp_sn_head")
        return scale
