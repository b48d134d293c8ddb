"""Parameter manifest, seeded synthetic weights and seeded synthetic input batches.

* ``param_manifest`` lists the state_dict names/shapes of the reference ``PoseNet``
  (network/PoseNet.py:135-171) in reference order.  Non-backbone names were read from
  the reference itself (tests/golden/state_dict_manifest.json pins them); ``backbone.*``
  follows timm 0.9.6 ``FeatureListNet(convnext_base)`` naming, which could not be
  verified here (timm is not installed) -- ``HF_TO_TIMM`` maps the HuggingFace
  ``ConvNextModel`` names used for the cross-check.
* ``synth_state_dict`` is the "random-init weights" of BASELINE.json: every tensor is
  drawn from its own numpy Philox stream keyed by (seed, crc32(name)), so the golden
  generator, the tests and bench.py reproduce identical weights without shipping them.
  The reference's own init (normal std=1e-3 everywhere, conv_pnp_net.py:124-134) makes
  the rot6d logits ~1e-4 and the normalisation ill-conditioned, so a
  variance-preserving init is used instead.
"""
import re
import zlib
from collections import OrderedDict

import numpy as np

from .config import PoseNetConfig

# REAL275 intrinsics (evaluation/load_data_eval.py:157) and category mean sizes in metres
# (evaluation/load_data_eval.py:384-439, /1000)
REAL_INTRINSICS = np.array([[591.0125, 0, 322.525], [0, 590.16775, 244.11084], [0, 0, 1]], dtype=np.float32)
MEAN_SIZES = np.array([[87, 220, 89], [165, 80, 165], [88, 128, 156], [68, 146, 72], [346, 200, 335],
                       [146, 83, 114]], dtype=np.float32) / 1000.0


def _convnext_manifest(cfg, m):
    dims, depths = cfg.convnext_dims, cfg.convnext_depths
    m["backbone.stem_0.weight"] = (dims[0], 3, 4, 4)
    m["backbone.stem_0.bias"] = (dims[0],)
    m["backbone.stem_1.weight"] = (dims[0],)
    m["backbone.stem_1.bias"] = (dims[0],)
    for s, (d, n) in enumerate(zip(dims, depths)):
        if s > 0:
            m[f"backbone.stages_{s}.downsample.0.weight"] = (dims[s - 1],)
            m[f"backbone.stages_{s}.downsample.0.bias"] = (dims[s - 1],)
            m[f"backbone.stages_{s}.downsample.1.weight"] = (d, dims[s - 1], 2, 2)
            m[f"backbone.stages_{s}.downsample.1.bias"] = (d,)
        for b in range(n):
            p = f"backbone.stages_{s}.blocks.{b}"
            m[p + ".gamma"] = (d,)
            m[p + ".conv_dw.weight"] = (d, 1, 7, 7)
            m[p + ".conv_dw.bias"] = (d,)
            m[p + ".norm.weight"] = (d,)
            m[p + ".norm.bias"] = (d,)
            m[p + ".mlp.fc1.weight"] = (4 * d, d)
            m[p + ".mlp.fc1.bias"] = (4 * d,)
            m[p + ".mlp.fc2.weight"] = (d, 4 * d)
            m[p + ".mlp.fc2.bias"] = (d,)


RESNET34_LAYERS = ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2))   # (planes, blocks, stride) network/resnet.py:167-176


def _bn_manifest(prefix, c, m):
    for k, shp in (("weight", (c,)), ("bias", (c,)), ("running_mean", (c,)), ("running_var", (c,)), ("num_batches_tracked", ())):
        m[f"{prefix}.{k}"] = shp


def _resnet34_manifest(m):
    """network/resnet.py:121-152 (ResNet(BasicBlock,[3,4,6,3]) without avgpool/fc)."""
    m["backbone.conv1.weight"] = (64, 3, 7, 7)
    _bn_manifest("backbone.bn1", 64, m)
    inpl = 64
    for li, (planes, blocks, stride) in enumerate(RESNET34_LAYERS, 1):
        for b in range(blocks):
            p = f"backbone.layer{li}.{b}"
            m[p + ".conv1.weight"] = (planes, inpl if b == 0 else planes, 3, 3)
            _bn_manifest(p + ".bn1", planes, m)
            m[p + ".conv2.weight"] = (planes, planes, 3, 3)
            _bn_manifest(p + ".bn2", planes, m)
            if b == 0 and (stride != 1 or inpl != planes):
                m[p + ".downsample.0.weight"] = (planes, inpl, 1, 1)
                _bn_manifest(p + ".downsample.1", planes, m)
        inpl = planes


def _xyz_head_manifest(prefix, in_dim, m):
    m[f"{prefix}.features.0.weight"] = (in_dim, 256, 3, 3)  # ConvTranspose2d [Cin, Cout, kh, kw]
    m[f"{prefix}.features.1.weight"] = (256,)
    m[f"{prefix}.features.1.bias"] = (256,)
    for i in (3, 4, 6, 7, 9, 10):
        m[f"{prefix}.features.{i}.conv.weight"] = (256, 256, 3, 3)
        for alias in ("norm", "gn"):  # ConvModule registers the norm twice (conv_module.py:181-183)
            m[f"{prefix}.features.{i}.{alias}.weight"] = (256,)
            m[f"{prefix}.features.{i}.{alias}.bias"] = (256,)
    m[f"{prefix}.out_layer.weight"] = (3, 256, 1, 1)
    m[f"{prefix}.out_layer.bias"] = (3,)


def param_manifest(cfg: PoseNetConfig = PoseNetConfig()):
    """OrderedDict name -> shape, reference state_dict order (backbone first)."""
    m = OrderedDict()
    if cfg.main_backbone == "convnext":
        _convnext_manifest(cfg, m)
    elif cfg.main_backbone == "resnet34":   # throughput variant, not wired by the reference
        _resnet34_manifest(m)
    else:
        raise ValueError(cfg.main_backbone)
    fc = cfg.feature_channel
    _xyz_head_manifest("xyz_nocs_head", fc, m)
    m["size_head.conv1.weight"] = (cfg.feat_ts, fc, 1)
    m["size_head.conv1.bias"] = (cfg.feat_ts,)
    m["size_head.conv2.weight"] = (cfg.size_head_out_dim, cfg.feat_ts, 1)
    m["size_head.conv2.bias"] = (cfg.size_head_out_dim,)
    m["size_head.bn1.weight"] = (cfg.feat_ts,)
    m["size_head.bn1.bias"] = (cfg.feat_ts,)
    m["size_head.bn1.running_mean"] = (cfg.feat_ts,)
    m["size_head.bn1.running_var"] = (cfg.feat_ts,)
    m["size_head.bn1.num_batches_tracked"] = ()
    if cfg.nocsmap_encoder == "att":
        # MAPTransformerEncoer (network/attention_pnp_net.py:126-157): own Parameter first, then children in
        # registration order (norm, patch_embed, block); timm 0.9.6 Block(dim=256, num_heads=8): qkv without bias
        p = "nocs_encoder."
        m[p + "pos_embed"] = (1, 64, 256)
        m[p + "norm.weight"] = (256,)
        m[p + "norm.bias"] = (256,)
        m[p + "patch_embed.proj.weight"] = (256, 3, 8, 8)
        m[p + "patch_embed.proj.bias"] = (256,)
        for i in range(3):
            q = f"{p}block.{i}."
            m[q + "norm1.weight"] = (256,)
            m[q + "norm1.bias"] = (256,)
            m[q + "attn.qkv.weight"] = (768, 256)
            m[q + "attn.proj.weight"] = (256, 256)
            m[q + "attn.proj.bias"] = (256,)
            m[q + "norm2.weight"] = (256,)
            m[q + "norm2.bias"] = (256,)
            m[q + "mlp.fc1.weight"] = (1024, 256)
            m[q + "mlp.fc1.bias"] = (1024,)
            m[q + "mlp.fc2.weight"] = (256, 1024)
            m[q + "mlp.fc2.bias"] = (256,)
    for li, i in enumerate((0, 3, 6) if cfg.nocsmap_encoder == "conv" else ()):
        cin = 3 if li == 0 else 256
        p = f"nocs_encoder.features.{i}"
        if cfg.use_dcn == "dcnv3":
            m[p + ".conv.weight"] = (256, cin, 1, 1)
            m[p + ".conv.bias"] = (256,)
            m[p + ".dcnv3.dw_conv.0.weight"] = (256, 1, 3, 3)
            m[p + ".dcnv3.dw_conv.0.bias"] = (256,)
            m[p + ".dcnv3.dw_conv.1.1.weight"] = (256,)
            m[p + ".dcnv3.dw_conv.1.1.bias"] = (256,)
            m[p + ".dcnv3.offset.weight"] = (72, 256)
            m[p + ".dcnv3.offset.bias"] = (72,)
            m[p + ".dcnv3.mask.weight"] = (36, 256)
            m[p + ".dcnv3.mask.bias"] = (36,)
            m[p + ".dcnv3.input_proj.weight"] = (256, 256)
            m[p + ".dcnv3.input_proj.bias"] = (256,)
            m[p + ".dcnv3.output_proj.weight"] = (256, 256)
            m[p + ".dcnv3.output_proj.bias"] = (256,)
            m[p + ".bn.weight"] = (256,)          # DCNv3_C.bn: present, unused (dcnv3.py:28,36)
            m[p + ".bn.bias"] = (256,)
            m[p + ".bn.running_mean"] = (256,)
            m[p + ".bn.running_var"] = (256,)
            m[p + ".bn.num_batches_tracked"] = ()
        else:
            m[p + ".weight"] = (256, cin, 3, 3)   # nn.Conv2d(..., bias=False) conv_pnp_net.py:258-272
        m[f"nocs_encoder.features.{i + 1}.weight"] = (256,)
        m[f"nocs_encoder.features.{i + 1}.bias"] = (256,)
    m["feat_reducer.weight"] = (256, fc, 1, 1)
    m["feat_reducer.bias"] = (256,)
    _xyz_head_manifest("xyz_deform_head", 512, m)
    for li, i in enumerate((0, 3, 6)):
        m[f"pnp_net.features.{i}.weight"] = (128, 5 if li == 0 else 128, 3, 3)
        m[f"pnp_net.features.{i + 1}.weight"] = (128,)
        m[f"pnp_net.features.{i + 1}.bias"] = (128,)
    for n, shp in (("fc1", (1024, 8192)), ("fc2", (256, 1024)), ("fc1_z", (1024, 8192)),
                   ("fc2_z", (256, 1024)), ("fc_z", (1, 256)), ("fc_r", (6, 256)), ("fc_t", (2, 256))):
        m[f"pnp_net.{n}.weight"] = shp
        m[f"pnp_net.{n}.bias"] = (shp[0],)
    return m


# ---- HuggingFace ConvNextModel name -> timm FeatureListNet name (cross-check only)
def hf_to_timm(name: str):
    """'model.encoder.stages.2.layers.5.pwconv1.weight' -> 'stages_2.blocks.5.mlp.fc1.weight'.
    Returns None for tensors timm's features_only net does not have (the pooler LayerNorm)."""
    name = re.sub(r"^model\.", "", name)
    if name.startswith("layernorm."):
        return None
    name = name.replace("embeddings.patch_embeddings.", "stem_0.").replace("embeddings.layernorm.", "stem_1.")
    name = re.sub(r"^encoder\.stages\.(\d+)\.downsampling_layer\.(\d)\.", r"stages_\1.downsample.\2.", name)
    name = re.sub(r"^encoder\.stages\.(\d+)\.layers\.(\d+)\.", r"stages_\1.blocks.\2.", name)
    for a, b in (("layer_scale_parameter", "gamma"), (".dwconv.", ".conv_dw."), (".layernorm.", ".norm."),
                 (".pwconv1.", ".mlp.fc1."), (".pwconv2.", ".mlp.fc2.")):
        name = name.replace(a, b)
    return name


def _rng(seed, name):
    return np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, zlib.crc32(name.encode())]))


def synth_tensor(name: str, shape, seed: int = 0) -> np.ndarray:
    """Seeded synthetic value of one state_dict tensor (float32; int64 for counters).
    Aliased ConvModule norms ('.gn.' / '.norm.') share one stream."""
    key = name.replace(".gn.", ".norm.")
    r = _rng(seed, key)
    shape = tuple(shape)
    if name.endswith("num_batches_tracked"):
        return np.zeros(shape, dtype=np.int64)
    if name.endswith("running_var"):
        return r.uniform(0.5, 1.5, shape).astype(np.float32)
    if name.endswith("running_mean"):
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if name.endswith(".gamma"):                      # ConvNeXt layer scale
        return r.uniform(0.1, 0.3, shape).astype(np.float32)
    if re.search(r"backbone\.layer\d\.\d+\.bn2\.weight$", name):   # keep 16 residual adds from blowing up
        return (0.25 + 0.05 * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 1:
        if name.endswith(".weight"):                 # norm scales
            return (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
        return (0.05 * r.standard_normal(shape)).astype(np.float32)   # biases
    n = r.standard_normal(shape).astype(np.float32)
    if re.search(r"features\.0\.weight$", name) and "xyz_" in name:   # ConvTranspose2d [Cin,Cout,3,3], stride 2
        fan_in = shape[0] * shape[2] * shape[3] / 4.0
    else:
        fan_in = float(np.prod(shape[1:]))
    gain = 1.4
    if name.endswith("dcnv3.offset.weight"):
        gain = 2.5      # |offset| ~ 1-3 px so the bilinear gather is exercised off-grid
    elif name.endswith("dcnv3.mask.weight"):
        gain = 1.5
    elif name.endswith("out_layer.weight"):
        gain = 0.5      # coordinate maps O(0.5)
    elif re.search(r"pnp_net\.fc_[rtz]\.weight$", name):
        gain = 1.0
    elif "conv_dw" in name or "dw_conv" in name:
        gain = 1.0
    return (gain / np.sqrt(fan_in) * n).astype(np.float32)


def synth_state_dict(cfg: PoseNetConfig = PoseNetConfig(), seed: int = 0):
    """OrderedDict name -> np.ndarray for every tensor of ``param_manifest(cfg)``."""
    return OrderedDict((k, synth_tensor(k, s, seed)) for k, s in param_manifest(cfg).items())


def synth_batch(B: int, seed: int = 0, img_size: int = 256, out_res: int = 64):
    """Seeded synthetic ``data`` dict of numpy arrays with the keys/shapes/dtypes the eval loader
    produces (evaluation/load_data_eval.py:361-378); SURVEY.md §8(d)."""
    r = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, 0xB47C4]))
    d = {}
    d["roi_img"] = r.standard_normal((B, 3, img_size, img_size), dtype=np.float32)
    d["roi_mask"] = (r.random((B, 1, img_size, img_size), dtype=np.float32) > 0.5).astype(np.float32)
    bbox_center = np.stack([r.uniform(100, 540, B), r.uniform(100, 380, B)], 1).astype(np.float32)
    wh = r.uniform(40, 200, (B, 2)).astype(np.float32)
    scale = 1.5 * wh.max(1)                                       # DZI_PAD_SCALE * max(w, h)
    # roi_coord_2d: normalised [-1,1] pixel grid of a 640x480 frame sampled NEAREST at out_res
    lin = (np.arange(out_res, dtype=np.float32) + 0.5) / out_res - 0.5
    xs = bbox_center[:, 0:1] + lin[None, :] * scale[:, None]
    ys = bbox_center[:, 1:2] + lin[None, :] * scale[:, None]
    gx = np.clip(np.floor(xs), 0, 639) / 639.0 * 2 - 1
    gy = np.clip(np.floor(ys), 0, 479) / 479.0 * 2 - 1
    d["roi_coord_2d"] = np.stack([np.broadcast_to(gx[:, None, :], (B, out_res, out_res)),
                                  np.broadcast_to(gy[:, :, None], (B, out_res, out_res))], 1).astype(np.float32)
    d["cam_K"] = np.broadcast_to(REAL_INTRINSICS, (B, 3, 3)).copy()
    d["roi_wh"] = wh
    d["bbox_center"] = bbox_center
    d["resize_ratio"] = (out_res / scale).astype(np.float32)     # load_data_eval.py:267,331
    d["mean_size"] = MEAN_SIZES[r.integers(0, 6, B)]
    return d


# ------------------------------------------------------------------------------------------------- Scale_net
# torchvision 0.15.2 mobilenet_v3_small "features" (third-party arithmetic, restated from the MobileNetV3 paper /
# torchvision's inverted-residual settings -- UNPINNED: torchvision is not installed here):
#   (in, kernel, expanded, out, squeeze-excitation, activation, stride)
MBV3S = ((16, 3, 16, 16, True, "RE", 2), (16, 3, 72, 24, False, "RE", 2), (24, 3, 88, 24, False, "RE", 1),
         (24, 5, 96, 40, True, "HS", 2), (40, 5, 240, 40, True, "HS", 1), (40, 5, 240, 40, True, "HS", 1),
         (40, 5, 120, 48, True, "HS", 1), (48, 5, 144, 48, True, "HS", 1), (48, 5, 288, 96, True, "HS", 2),
         (96, 5, 576, 96, True, "HS", 1), (96, 5, 576, 96, True, "HS", 1))
MBV3S_LAST = 576


def make_divisible(v, divisor=8):
    """torchvision.ops.misc / _utils._make_divisible: squeeze channels of the SE block = make_divisible(expanded // 4, 8)."""
    new_v = max(divisor, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def scale_net_manifest(feat_dim=24, cats_num=6, use_hw=True):
    """state_dict names / shapes of the reference ``Scale_net`` (network/scale_net.py:22-43): two
    nn.Sequential(mobilenet_v3_small.features, avgpool, Flatten) encoders + line1..3.  The torchvision key layout
    (features.{i}.block.{j}.{0,1} = conv / BatchNorm, SqueezeExcitation fc1 / fc2) is from memory."""
    m = OrderedDict()
    for enc in ("feat_encoder_bbox", "feat_encoder_full"):
        f = enc + ".0"
        m[f + ".0.0.weight"] = (16, 3, 3, 3)
        _bn_manifest(f + ".0.1", 16, m)
        for i, (cin, k, exp, cout, se, _, _) in enumerate(MBV3S, 1):
            j = 0
            if exp != cin:
                m[f"{f}.{i}.block.{j}.0.weight"] = (exp, cin, 1, 1)
                _bn_manifest(f"{f}.{i}.block.{j}.1", exp, m)
                j += 1
            m[f"{f}.{i}.block.{j}.0.weight"] = (exp, 1, k, k)
            _bn_manifest(f"{f}.{i}.block.{j}.1", exp, m)
            j += 1
            if se:
                sq = make_divisible(exp // 4, 8)
                m[f"{f}.{i}.block.{j}.fc1.weight"] = (sq, exp, 1, 1)
                m[f"{f}.{i}.block.{j}.fc1.bias"] = (sq,)
                m[f"{f}.{i}.block.{j}.fc2.weight"] = (exp, sq, 1, 1)
                m[f"{f}.{i}.block.{j}.fc2.bias"] = (exp,)
                j += 1
            m[f"{f}.{i}.block.{j}.0.weight"] = (cout, exp, 1, 1)
            _bn_manifest(f"{f}.{i}.block.{j}.1", cout, m)
        m[f + ".12.0.weight"] = (MBV3S_LAST, 96, 1, 1)
        _bn_manifest(f + ".12.1", MBV3S_LAST, m)
    m["line1.weight"], m["line1.bias"] = (128, 2 * MBV3S_LAST), (128,)
    m["line2.weight"], m["line2.bias"] = (feat_dim, 128 + cats_num), (feat_dim,)
    m["line3.weight"], m["line3.bias"] = (1, feat_dim + (2 if use_hw else 0) + cats_num), (1,)
    return m


def synth_scale_net_state_dict(feat_dim=24, seed=0):
    return OrderedDict((k, synth_tensor("scale_net." + k, s, seed)) for k, s in scale_net_manifest(feat_dim).items())


def synth_scale_batch(B, seed=0, img_size=256):
    """Extra eval-loader keys Scale_net reads (evaluation/load_data_eval.py:336-361, evaluate.py:101-102)."""
    r = np.random.Generator(np.random.Philox(key=[seed & 0xFFFFFFFF, 0x5CA1E]))
    d = synth_batch(B, seed, img_size)
    d["full_img"] = np.broadcast_to(r.standard_normal((1, 3, img_size, img_size), dtype=np.float32), (B, 3, img_size, img_size)).copy()
    cat = r.integers(0, 6, B)
    d["one_hot"] = np.eye(6, dtype=np.float32)[cat]
    return d
