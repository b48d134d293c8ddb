"""Import shim for running the GIVEPose reference IN THE BUILD CONTAINER ONLY.

Purpose: generate golden vectors (scripts/gen_golden.py) and cross-check the oracle
(oracle/posenet_ref.py) against the real reference modules under /root/reference.
Nothing here is imported by the product path, the tests or bench.py, and nothing
here travels as reference code: it only *stubs the third-party packages the
reference imports but this image lacks* (absl, mmcv, timm, torchvision,
detectron2, cv2, ipdb, open3d, transforms3d, the compiled DCNv3 extension).

Stubs that carry arithmetic restate the published behaviour of the pinned
third-party version (GIVEPose_env.yml):
  * transforms3d 0.4.1 ``axangle2mat``  -> Rodrigues formula
  * torchvision 0.15.2 ``Resize(NEAREST)`` on tensors -> F.interpolate(mode='nearest')
  * mmcv 1.5.0 ``normal_init/constant_init/kaiming_init`` -> nn.init wrappers
  * timm 0.9.6 ``convnext_base(features_only, out_indices=(3,))`` -> HF
    ``transformers.ConvNextModel`` stand-in with the same architecture (timm is
    not installed; backbone parity is therefore pinned against HF)
  * DCNv3.dcnv3_forward (CUDA only in the reference) -> the reference's own
    ``dcnv3_core_pytorch`` applied to the buffer prefix the CUDA kernel actually
    consumes (dcnv3_cuda.cu:40-65, dcnv3_im2col_cuda.cuh:226-244).
"""
import enum
import importlib.abc
import importlib.machinery
import math
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = "/root/reference"

_STUB_TOPLEVEL = {
    "absl", "mmcv", "timm", "torchvision", "detectron2", "cv2", "ipdb",
    "open3d", "transforms3d", "tensorboardX", "tensorboard", "DCNv3", "termcolor",
    "skimage", "imageio", "pycocotools", "trimesh", "pyrender", "ruamel", "pytorch3d", "fvcore",
}


_SUBMODULE_NAMES = {
    "flags", "app", "cnn", "utils", "bricks", "conv", "padding", "models", "layers", "registry",
    "vision_transformer", "transforms", "axangles", "quaternions", "euler", "batch_norm", "env",
}


class _Dummy:
    """Placeholder for names that are imported but never used on the path."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __iter__(self):
        return iter(())

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = self.__name__ + "." + name
        if full in sys.modules:
            return sys.modules[full]
        if name in _SUBMODULE_NAMES:
            import importlib
            return importlib.import_module(full)
        return _Dummy()


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path, target=None):
        if fullname.split(".")[0] in _STUB_TOPLEVEL:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _StubModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        _populate(module)


# --------------------------------------------------------------------------- absl
class _Flags:
    def __init__(self):
        object.__setattr__(self, "_v", {})

    def __getattr__(self, k):
        try:
            return self._v[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self._v[k] = v

    def __contains__(self, k):
        return k in self._v


FLAGS = _Flags()


def _define(name, default, help=None, **kw):
    if name not in FLAGS:
        setattr(FLAGS, name, default)


# --------------------------------------------------------------------------- mmcv
def _normal_init(module, mean=0, std=1, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.normal_(module.weight, mean, std)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def _constant_init(module, val, bias=0):
    if hasattr(module, "weight") and module.weight is not None:
        nn.init.constant_(module.weight, val)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


def _kaiming_init(module, a=0, mode="fan_out", nonlinearity="relu", bias=0, distribution="normal"):
    if hasattr(module, "weight") and module.weight is not None:
        if distribution == "uniform":
            nn.init.kaiming_uniform_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
        else:
            nn.init.kaiming_normal_(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if hasattr(module, "bias") and module.bias is not None:
        nn.init.constant_(module.bias, bias)


class _Registry:
    def __init__(self):
        self._d = {"Conv1d": nn.Conv1d, "Conv2d": nn.Conv2d, "Conv3d": nn.Conv3d, "Conv": nn.Conv2d}

    def get(self, k):
        return self._d.get(k)

    def __contains__(self, k):
        return k in self._d

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._d[name or module.__name__] = module
            return module

        def deco(cls):
            self._d[name or cls.__name__] = cls
            return cls

        return deco


_CONV_LAYERS = _Registry()


def _build_conv_layer(cfg, *args, **kwargs):
    cfg_ = dict(type="Conv2d") if cfg is None else dict(cfg)
    t = cfg_.pop("type")
    return _CONV_LAYERS.get(t)(*args, **kwargs, **cfg_)


# --------------------------------------------------------------------------- timm
class _StdConv2d(nn.Conv2d):
    def __init__(self, in_channel, out_channels, kernel_size, stride=1, padding=None,
                 dilation=1, groups=1, bias=False, eps=1e-6):
        if padding is None:
            padding = ((stride - 1) + dilation * (kernel_size - 1)) // 2
        super().__init__(in_channel, out_channels, kernel_size, stride=stride, padding=padding,
                         dilation=dilation, groups=groups, bias=bias)
        self.eps = eps

    def forward(self, x):
        w = F.batch_norm(self.weight.reshape(1, self.out_channels, -1), None, None,
                         training=True, momentum=0., eps=self.eps).reshape_as(self.weight)
        return F.conv2d(x, w, self.bias, self.stride, self.padding, self.dilation, self.groups)


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob

    def forward(self, x):
        assert not (self.training and self.drop_prob > 0)
        return x


class _Mlp(nn.Module):
    """timm 0.9.6 Mlp: fc1 -> act -> drop -> fc2 -> drop."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU,
                 norm_layer=None, bias=True, drop=0.0, use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class _Attention(nn.Module):
    """timm 0.9.6 vision_transformer.Attention (qkv packed, scale = head_dim**-0.5)."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_norm=False, attn_drop=0., proj_drop=0.,
                 norm_layer=nn.LayerNorm):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = nn.Identity()
        self.k_norm = nn.Identity()
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv.unbind(0)
        attn = (q * self.scale) @ k.transpose(-2, -1)
        attn = attn.softmax(dim=-1)
        x = (attn @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj_drop(self.proj(x))


class _Block(nn.Module):
    """timm 0.9.6 vision_transformer.Block: pre-norm MHA + MLP, LayerNorm eps 1e-5 default."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, qk_norm=False, proj_drop=0.,
                 attn_drop=0., init_values=None, drop_path=0., act_layer=nn.GELU,
                 norm_layer=nn.LayerNorm, mlp_layer=None, drop=0.0, **kw):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = _Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias, attn_drop=attn_drop,
                               proj_drop=proj_drop or drop)
        self.ls1 = nn.Identity()
        self.drop_path1 = nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = _Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer,
                        drop=proj_drop or drop)
        self.ls2 = nn.Identity()
        self.drop_path2 = nn.Identity()

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x


class _HFConvNeXtFeatures(nn.Module):
    """Stand-in for timm FeatureListNet(convnext_base, out_indices=(3,)): returns [stage-4 map]."""

    def __init__(self):
        super().__init__()
        from transformers import ConvNextConfig, ConvNextModel
        cfg = ConvNextConfig(num_channels=3, hidden_sizes=[128, 256, 512, 1024], depths=[3, 3, 27, 3],
                             layer_scale_init_value=1e-6, drop_path_rate=0.0)
        self.model = ConvNextModel(cfg)
        self.default_cfg = {}

    def forward(self, x):
        out = self.model(pixel_values=x, return_dict=True)
        return [out.last_hidden_state]


def _timm_create_model(model_name="convnext_base", **kw):
    assert model_name == "convnext_base", model_name
    return _HFConvNeXtFeatures()


# --------------------------------------------------------------------------- torchvision
class _InterpolationMode(enum.Enum):
    NEAREST = "nearest"
    BILINEAR = "bilinear"
    BICUBIC = "bicubic"


class _Resize(nn.Module):
    def __init__(self, size, interpolation=_InterpolationMode.BILINEAR, max_size=None, antialias=None):
        super().__init__()
        self.size = size if isinstance(size, (tuple, list)) else (size, size)
        self.interpolation = interpolation

    def forward(self, img):
        assert self.interpolation == _InterpolationMode.NEAREST
        assert img.shape[-1] == img.shape[-2]  # int size == shorter side; square crops only
        return F.interpolate(img, size=tuple(self.size), mode="nearest")


class _ConvBNAct(nn.Sequential):
    """torchvision.ops.misc.Conv2dNormActivation: [Conv2d(bias=False), BatchNorm2d(eps 1e-3, momentum 1e-2), act] [memory]."""

    def __init__(self, cin, cout, k=1, stride=1, groups=1, act=None):
        layers = [nn.Conv2d(cin, cout, k, stride, k // 2, groups=groups, bias=False), nn.BatchNorm2d(cout, eps=1e-3, momentum=1e-2)]
        if act is not None:
            layers.append(act())
        super().__init__(*layers)
        self.out_channels = cout


class _SqueezeExcitation(nn.Module):
    """torchvision.ops.misc.SqueezeExcitation(activation=ReLU, scale_activation=Hardsigmoid) [memory]."""

    def __init__(self, c, sq):
        super().__init__()
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1, self.fc2 = nn.Conv2d(c, sq, 1), nn.Conv2d(sq, c, 1)
        self.activation, self.scale_activation = nn.ReLU(), nn.Hardsigmoid()

    def forward(self, x):
        return x * self.scale_activation(self.fc2(self.activation(self.fc1(self.avgpool(x)))))


class _InvertedResidual(nn.Module):
    """torchvision.models.mobilenetv3.InvertedResidual [memory]."""

    def __init__(self, cin, k, exp, cout, se, act, stride):
        super().__init__()
        from givepose_amd.synth import make_divisible
        A = nn.Hardswish if act == "HS" else nn.ReLU
        layers = []
        if exp != cin:
            layers.append(_ConvBNAct(cin, exp, 1, act=A))
        layers.append(_ConvBNAct(exp, exp, k, stride, groups=exp, act=A))
        if se:
            layers.append(_SqueezeExcitation(exp, make_divisible(exp // 4, 8)))
        layers.append(_ConvBNAct(exp, cout, 1, act=None))
        self.block = nn.Sequential(*layers)
        self.use_res_connect = stride == 1 and cin == cout

    def forward(self, x):
        y = self.block(x)
        return x + y if self.use_res_connect else y


class _MobileNetV3Small(nn.Module):
    """Stand-in for torchvision.models.mobilenet_v3_small (0.15.2), restated from memory / the MobileNetV3 paper: only
    ``features`` and ``avgpool`` are used by network/scale_net.py:25-29.  Pins the Scale_net WIRING, not torchvision."""

    def __init__(self):
        super().__init__()
        from givepose_amd.synth import MBV3S, MBV3S_LAST
        layers = [_ConvBNAct(3, 16, 3, 2, act=nn.Hardswish)]
        layers += [_InvertedResidual(*cfg) for cfg in MBV3S]
        layers.append(_ConvBNAct(96, MBV3S_LAST, 1, act=nn.Hardswish))
        self.features = nn.Sequential(*layers)
        self.avgpool = nn.AdaptiveAvgPool2d(1)


def _mobilenet_v3_small(pretrained=False, **kw):
    return _MobileNetV3Small()


# --------------------------------------------------------------------------- transforms3d
def _axangle2mat(axis, angle, is_normalized=False):
    x, y, z = axis
    if not is_normalized:
        n = math.sqrt(x * x + y * y + z * z)
        x, y, z = x / n, y / n, z / n
    c, s = math.cos(angle), math.sin(angle)
    C = 1 - c
    xs, ys, zs = x * s, y * s, z * s
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    return np.array([[x * xC + c, xyC - zs, zxC + ys],
                     [xyC + zs, y * yC + c, yzC - xs],
                     [zxC - ys, yzC + xs, z * zC + c]])


# --------------------------------------------------------------------------- DCNv3 extension
def _dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                   dilation_h, dilation_w, group, group_channels, offset_scale, im2col_step,
                   remove_center=0):
    """CPU stand-in for the CUDA op.  The CUDA kernel indexes offset/mask linearly by
    ((b*Ho+ho)*Wo+wo)*G+g (dcnv3_im2col_cuda.cuh:226-244), i.e. it consumes the first
    N*Ho*Wo rows of the flat (.., G*P*2)/(.., G*P) buffers whatever their nominal shape."""
    from network.ops_dcnv3.functions.dcnv3_func import dcnv3_core_pytorch
    N, H, W, _ = input.shape
    Ho = (H + 2 * pad_h - (dilation_h * (kernel_h - 1) + 1)) // stride_h + 1
    Wo = (W + 2 * pad_w - (dilation_w * (kernel_w - 1) + 1)) // stride_w + 1
    P = kernel_h * kernel_w - remove_center
    assert N <= im2col_step or N % im2col_step == 0
    off = offset.reshape(-1, group * P * 2)[: N * Ho * Wo].reshape(N, Ho, Wo, group * P * 2)
    msk = mask.reshape(-1, group * P)[: N * Ho * Wo].reshape(N, Ho, Wo, group * P)
    return dcnv3_core_pytorch(input, off, msk, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                              dilation_h, dilation_w, group, group_channels, offset_scale,
                              remove_center)


# --------------------------------------------------------------------------- wiring
def _populate(m):
    n = m.__name__
    if n == "absl":
        pass
    elif n == "absl.flags":
        m.FLAGS = FLAGS
        for k in ("DEFINE_string", "DEFINE_integer", "DEFINE_float", "DEFINE_bool", "DEFINE_boolean",
                  "DEFINE_list", "DEFINE_enum"):
            setattr(m, k, _define)
    elif n in ("mmcv.cnn", "mmcv.cnn.utils"):
        m.normal_init, m.constant_init, m.kaiming_init = _normal_init, _constant_init, _kaiming_init
    elif n == "mmcv.cnn.bricks.conv":
        m.CONV_LAYERS = _CONV_LAYERS
        m.build_conv_layer = _build_conv_layer
    elif n in ("timm.models.layers", "timm.layers"):
        m.StdConv2d = _StdConv2d
        m.trunc_normal_ = nn.init.trunc_normal_
        m.DropPath = _DropPath
        m.to_2tuple = lambda x: x if isinstance(x, (tuple, list)) else (x, x)
        m.Mlp = _Mlp
    elif n == "timm.models.registry":
        m.register_model = lambda f: f
    elif n == "timm.models.vision_transformer":
        m.Block, m.Mlp, m.Attention = _Block, _Mlp, _Attention
        m._cfg = lambda **k: {}
    elif n == "timm":
        m.create_model = _timm_create_model
    elif n == "torchvision.models":
        m.mobilenet_v3_small = _mobilenet_v3_small
    elif n == "torchvision.transforms":
        m.Resize = _Resize
        m.InterpolationMode = _InterpolationMode
    elif n == "transforms3d.axangles":
        m.axangle2mat = _axangle2mat
    elif n == "detectron2.layers.batch_norm":
        m.BatchNorm2d = nn.BatchNorm2d
        m.FrozenBatchNorm2d = type("FrozenBatchNorm2d", (nn.Module,), {})
        m.NaiveSyncBatchNorm = type("NaiveSyncBatchNorm", (nn.BatchNorm2d,), {})
    elif n == "detectron2.utils":
        pass
    elif n == "detectron2.utils.env":
        m.TORCH_VERSION = (2, 0)
    elif n == "DCNv3":
        m.dcnv3_forward = _dcnv3_forward


_installed = False


def install():
    """Install the stubs and put the reference on sys.path. Idempotent."""
    global _installed
    if _installed:
        return FLAGS
    _installed = True
    sys.dont_write_bytecode = True
    # GP_SHIM_REAL=timm,torchvision,cv2: use the INSTALLED package instead of this file's stand-in for the names listed (re-verifying
    # rows a2 / a13 / f2 / f1 where the packages exist).  Default: every name of _STUB_TOPLEVEL is stubbed, installed or not (the committed vectors were made that way).
    import importlib.util
    import os
    for name in filter(None, os.environ.get("GP_SHIM_REAL", "").split(",")):
        if name in _STUB_TOPLEVEL and importlib.util.find_spec(name) is not None:
            _STUB_TOPLEVEL.discard(name)
            print(f"[ref_shim] using the installed {name}", file=sys.stderr)
        else:
            print(f"[ref_shim] GP_SHIM_REAL: {name} is not installed (or not a stubbed name): keeping the stand-in", file=sys.stderr)
    # transformers probes for torchvision at import time: load it before the stubs exist
    from transformers import ConvNextConfig, ConvNextModel  # noqa: F401
    sys.meta_path.insert(0, _StubFinder())
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    if not hasattr(np, "maximum_sctype"):  # removed in NumPy 2; RT_transform.py:297
        np.maximum_sctype = lambda t: np.float64
    import pkg_resources
    _orig = pkg_resources.get_distribution

    def _get_distribution(name):
        if name == "DCNv3":
            return types.SimpleNamespace(version="1.1")
        return _orig(name)

    pkg_resources.get_distribution = _get_distribution
    import config.config  # noqa: F401  (reference flag definitions -> FLAGS defaults)
    return FLAGS
