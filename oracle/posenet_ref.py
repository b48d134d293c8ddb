"""ORACLE -- test infrastructure, not product code.

CPU restatement (plain PyTorch fp32/fp64 + numpy) of the GIVEPose ``PoseNet.forward``
inference path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file; the product (givepose_amd/) never does.

Pinned by: tests/golden/*.npz, generated in the build container by
scripts/gen_golden.py from an import of the reference modules themselves (same seeded
weights, same inputs), which this restatement matches to <=2e-5 (see
tests/test_oracle_golden.py).  The ConvNeXt trunk is third-party arithmetic (timm 0.9.6,
not vendored, not installed): it is pinned against HuggingFace ``ConvNextModel`` of the
same architecture instead -- parity against timm itself is UNPINNED.

Every function cites the reference file:line it follows (paths relative to the
reference root).  Parameters are passed as a flat dict ``name -> tensor`` using the
reference's state_dict names.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- DCNv3 core
def dcnv3_forward_ref(inp, offset, mask, K, stride, pad, dil, G, D, offset_scale, remove_center=0):
    """network/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:216-282 (+ bilinear :32-80) and the host
    wrapper dcnv3_cuda.cu:21-85, vectorised.  ``offset``/``mask`` are consumed as FLAT buffers
    indexed by ((b*Ho+ho)*Wo+wo)*G+g -- whatever their nominal shape -- exactly as the kernel
    does (:238-244).  inp: (N,H,W,G*D) channels-last.  Returns (N,Ho,Wo,G*D)."""
    N, H, W, C = inp.shape
    assert C == G * D
    Ho = (H + 2 * pad - (dil * (K - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (K - 1) + 1)) // stride + 1
    P = K * K - remove_center
    dt = inp.dtype
    off = offset.reshape(-1)[: N * Ho * Wo * G * P * 2].reshape(N, Ho, Wo, G, P, 2).to(dt)
    msk = mask.reshape(-1)[: N * Ho * Wo * G * P].reshape(N, Ho, Wo, G, P).to(dt)
    half = (dil * (K - 1)) >> 1
    wo = torch.arange(Wo, dtype=dt).view(1, 1, Wo, 1)
    ho = torch.arange(Ho, dtype=dt).view(1, Ho, 1, 1)
    p0_w_ = (half - pad + wo * stride) - half * offset_scale      # :246-249
    p0_h_ = (half - pad + ho * stride) - half * offset_scale
    x = inp.reshape(N, H * W, G, D)
    out = torch.zeros(N, Ho, Wo, G, D, dtype=dt)
    bidx = torch.arange(N).view(N, 1, 1, 1).expand(N, Ho, Wo, G)
    gidx = torch.arange(G).view(1, 1, 1, G).expand(N, Ho, Wo, G)
    p = 0
    for i in range(K):            # kernel_w outer (:254)
        for j in range(K):        # kernel_h inner (:255)
            if remove_center and i == K // 2 and j == K // 2:
                continue
            loc_w = p0_w_ + (i * dil + off[..., p, 0]) * offset_scale   # :259-262
            loc_h = p0_h_ + (j * dil + off[..., p, 1]) * offset_scale
            valid = (loc_h > -1) & (loc_w > -1) & (loc_h < H) & (loc_w < W)  # :264-265
            h_low = torch.floor(loc_h)
            w_low = torch.floor(loc_w)
            lh, lw = loc_h - h_low, loc_w - w_low
            hh, hw = 1 - lh, 1 - lw
            h_low, w_low = h_low.long(), w_low.long()
            acc = torch.zeros(N, Ho, Wo, G, D, dtype=dt)
            for (hy, wx, wt) in ((h_low, w_low, hh * hw), (h_low, w_low + 1, hh * lw),
                                 (h_low + 1, w_low, lh * hw), (h_low + 1, w_low + 1, lh * lw)):
                ok = valid & (hy >= 0) & (hy <= H - 1) & (wx >= 0) & (wx <= W - 1)   # :51-70
                lin = (hy.clamp(0, H - 1) * W + wx.clamp(0, W - 1))
                v = x[bidx, lin, gidx]                                              # (N,Ho,Wo,G,D)
                acc = acc + (wt * ok.to(dt)).unsqueeze(-1) * v
            out = out + acc * msk[..., p].unsqueeze(-1)
            p += 1
    return out.reshape(N, Ho, Wo, G * D)


# ----------------------------------------------------------------------------- small layers
def _gn(x, w, b, groups=32):
    return F.group_norm(x, groups, w, b, eps=1e-5)


def _ln_cl(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def convnext_ref(P, img, cfg, prefix="backbone."):
    """timm 0.9.6 convnext_base FeatureListNet(out_indices=(3,)) as called by
    network/backbone.py:36-46; arithmetic per the ConvNeXt paper / timm ConvNeXtBlock:
    stem conv4x4 s4 + LayerNorm2d(eps 1e-6); per stage [LayerNorm2d + conv2x2 s2] then
    blocks x + gamma * fc2(GELU(fc1(LN(dw7x7(x))))).  Returns [ (B,1024,8,8) ]."""
    g = lambda k: P[prefix + k]
    x = F.conv2d(img, g("stem_0.weight"), g("stem_0.bias"), stride=4)
    x = _ln_cl(x.permute(0, 2, 3, 1), g("stem_1.weight"), g("stem_1.bias"), 1e-6).permute(0, 3, 1, 2)
    for s, n in enumerate(cfg.convnext_depths):
        if s > 0:
            x = _ln_cl(x.permute(0, 2, 3, 1), g(f"stages_{s}.downsample.0.weight"),
                       g(f"stages_{s}.downsample.0.bias"), 1e-6).permute(0, 3, 1, 2)
            x = F.conv2d(x, g(f"stages_{s}.downsample.1.weight"), g(f"stages_{s}.downsample.1.bias"), stride=2)
        for b in range(n):
            p = f"stages_{s}.blocks.{b}."
            y = F.conv2d(x, g(p + "conv_dw.weight"), g(p + "conv_dw.bias"), padding=3, groups=x.shape[1])
            y = _ln_cl(y.permute(0, 2, 3, 1), g(p + "norm.weight"), g(p + "norm.bias"), 1e-6)
            y = F.linear(y, g(p + "mlp.fc1.weight"), g(p + "mlp.fc1.bias"))
            y = F.gelu(y)
            y = F.linear(y, g(p + "mlp.fc2.weight"), g(p + "mlp.fc2.bias"))
            x = x + (g(p + "gamma") * y).permute(0, 3, 1, 2)
    return [x]


def resnet34_ref(P, img, prefix="backbone."):
    """network/resnet.py:137-147 (ResNet.forward up to layer4) with BasicBlock :24-52, eval BatchNorm.
    Returns [ (B,512,H/32,W/32) ]."""
    def bn(x, p):
        return F.batch_norm(x, P[p + ".running_mean"], P[p + ".running_var"], P[p + ".weight"], P[p + ".bias"], False, 0.0, 1e-5)
    x = F.relu(bn(F.conv2d(img, P[prefix + "conv1.weight"], None, stride=2, padding=3), prefix + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    for li, (planes, blocks, stride) in enumerate(((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)), 1):
        for b in range(blocks):
            p = f"{prefix}layer{li}.{b}"
            s = stride if b == 0 else 1
            out = F.relu(bn(F.conv2d(x, P[p + ".conv1.weight"], None, stride=s, padding=1), p + ".bn1"))
            out = bn(F.conv2d(out, P[p + ".conv2.weight"], None, padding=1), p + ".bn2")
            res = x
            if (p + ".downsample.0.weight") in P:
                res = bn(F.conv2d(x, P[p + ".downsample.0.weight"], None, stride=s), p + ".downsample.1")
            x = F.relu(out + res)
    return [x]


def size_head_ref(P, feat, prefix="size_head."):
    """network/pose_head.py:30-42 (eval: BN uses running stats, dropout is identity)."""
    x = feat.flatten(2, 3).max(dim=-1, keepdim=True).values
    x = F.conv1d(x, P[prefix + "conv1.weight"], P[prefix + "conv1.bias"])
    x = F.batch_norm(x, P[prefix + "bn1.running_mean"], P[prefix + "bn1.running_var"],
                     P[prefix + "bn1.weight"], P[prefix + "bn1.bias"], False, 0.0, 1e-5)
    x = F.relu(x)
    x = F.conv1d(x, P[prefix + "conv2.weight"], P[prefix + "conv2.bias"])
    return x.squeeze(2)[:, :3]


def xyz_head_ref(P, x, prefix):
    """network/xyz_head.py:349-366, layers built at :241-316 with the defaults
    up_types=(deconv,bilinear,bilinear), 2 convs per block, GN(32), GELU, 1x1 out layer."""
    g = lambda k: P[prefix + k]
    x = F.conv_transpose2d(x, g("features.0.weight"), None, stride=2, padding=1, output_padding=1)
    x = F.gelu(_gn(x, g("features.1.weight"), g("features.1.bias")))
    for i in (3, 4, 6, 7, 9, 10):
        if i in (6, 9):   # features.5 / features.8 = UpsamplingBilinear2d(scale 2) -> align_corners=True
            x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
        x = F.conv2d(x, g(f"features.{i}.conv.weight"), None, padding=1)
        x = F.gelu(_gn(x, g(f"features.{i}.norm.weight"), g(f"features.{i}.norm.bias")))
    out = F.conv2d(x, g("out_layer.weight"), g("out_layer.bias"))
    return out  # (B,3,64,64) == cat(coor_x, coor_y, coor_z) of :357-362


def dcnv3_module_ref(P, x_nhwc, prefix, stride=2, G=4, K=3, pad=1, dil=1, offset_scale=1.0):
    """network/ops_dcnv3/modules/dcnv3.py:318-356 (DCNv3.forward).  offset/mask are produced at
    FULL resolution (stride-1 depth-wise conv) and handed flat to the core, which only consumes
    the first N*Ho*Wo rows."""
    g = lambda k: P[prefix + k]
    N, H, W, C = x_nhwc.shape
    x = F.linear(x_nhwc, g("input_proj.weight"), g("input_proj.bias"))
    x1 = F.conv2d(x_nhwc.permute(0, 3, 1, 2), g("dw_conv.0.weight"), g("dw_conv.0.bias"), padding=1, groups=C)
    x1 = F.gelu(_ln_cl(x1.permute(0, 2, 3, 1), g("dw_conv.1.1.weight"), g("dw_conv.1.1.bias"), 1e-6))
    offset = F.linear(x1, g("offset.weight"), g("offset.bias"))
    mask = F.linear(x1, g("mask.weight"), g("mask.bias")).reshape(N, H, W, G, -1)
    mask = F.softmax(mask, -1).reshape(N, H, W, -1)
    y = dcnv3_forward_ref(x, offset, mask, K, stride, pad, dil, G, C // G, offset_scale, 0)
    return F.linear(y, g("output_proj.weight"), g("output_proj.bias"))


def map_transformer_ref(P, coor, prefix="nocs_encoder."):
    """network/attention_pnp_net.py:143-157 (MAPTransformerEncoer.forward), PatchEmbed :297-302, and timm 0.9.6
    vision_transformer.Block / Attention / Mlp [third-party, from memory -- UNPINNED against timm]: pre-norm MHA
    (8 heads x 32, qkv without bias, scale 32**-0.5) + MLP (x4, exact GELU), LayerNorm eps 1e-5.  (B,3,64,64)->(B,256,8,8)."""
    g = lambda k: P[prefix + k]
    x = F.conv2d(coor, g("patch_embed.proj.weight"), g("patch_embed.proj.bias"), stride=8).flatten(2).transpose(1, 2)
    x = x + g("pos_embed")
    B, N, C = x.shape
    for i in range(3):
        q = f"block.{i}."
        h = F.layer_norm(x, (C,), g(q + "norm1.weight"), g(q + "norm1.bias"), 1e-5)
        qkv = F.linear(h, g(q + "attn.qkv.weight")).reshape(B, N, 3, 8, C // 8).permute(2, 0, 3, 1, 4)
        qq, kk, vv = qkv.unbind(0)
        att = ((qq * (C // 8) ** -0.5) @ kk.transpose(-2, -1)).softmax(dim=-1)
        h = (att @ vv).transpose(1, 2).reshape(B, N, C)
        x = x + F.linear(h, g(q + "attn.proj.weight"), g(q + "attn.proj.bias"))
        h = F.layer_norm(x, (C,), g(q + "norm2.weight"), g(q + "norm2.bias"), 1e-5)
        h = F.linear(F.gelu(F.linear(h, g(q + "mlp.fc1.weight"), g(q + "mlp.fc1.bias"))), g(q + "mlp.fc2.weight"), g(q + "mlp.fc2.bias"))
        x = x + h
    x = F.layer_norm(x, (C,), g("norm.weight"), g("norm.bias"), 1e-5)
    return x.permute(0, 2, 1).reshape(B, C, 8, 8)


def map_encoder_ref(P, coor, cfg, prefix="nocs_encoder."):
    """network/conv_pnp_net.py:303-332 (MAPEncoder.forward) with layers :254-272:
    3 x [DCNv3_C(s2) | Conv3x3 s2 (use_dcn='')] -> GN(32) -> ReLU.  DCNv3_C: network/dcnv3.py:32-38."""
    x = coor
    for i in (0, 3, 6):
        p = f"{prefix}features.{i}."
        if cfg.use_dcn == "dcnv3":
            x = F.conv2d(x, P[p + "conv.weight"], P[p + "conv.bias"])
            x = dcnv3_module_ref(P, x.permute(0, 2, 3, 1), p + "dcnv3.").permute(0, 3, 1, 2)
        else:
            x = F.conv2d(x, P[p + "weight"], None, stride=2, padding=1)
        x = F.relu(_gn(x, P[f"{prefix}features.{i + 1}.weight"], P[f"{prefix}features.{i + 1}.bias"]))
    return x


def conv_pnp_ref(P, x, prefix="pnp_net."):
    """network/conv_pnp_net.py:137-201 (ConvPnPNet.forward; mask_attention 'none', flat_op 'flatten').
    Returns rot (B,6), t (B,3)."""
    g = lambda k: P[prefix + k]
    for i in (0, 3, 6):
        x = F.conv2d(x, g(f"features.{i}.weight"), None, stride=2, padding=1)
        x = F.relu(_gn(x, g(f"features.{i + 1}.weight"), g(f"features.{i + 1}.bias")))
    flat = x.flatten(1)
    act = lambda v: F.leaky_relu(v, 0.1)
    h = act(F.linear(flat, g("fc1.weight"), g("fc1.bias")))
    h = act(F.linear(h, g("fc2.weight"), g("fc2.bias")))
    rot = F.linear(h, g("fc_r.weight"), g("fc_r.bias"))
    t = F.linear(h, g("fc_t.weight"), g("fc_t.bias"))
    hz = act(F.linear(flat, g("fc1_z.weight"), g("fc1_z.bias")))
    hz = act(F.linear(hz, g("fc2_z.weight"), g("fc2_z.bias")))
    z = F.linear(hz, g("fc_z.weight"), g("fc_z.bias"))
    return rot, torch.cat([t, z], dim=1)


def rot6d_to_mat_ref(d6):
    """network/pose_utils/rot_reps.py:34-55."""
    x = F.normalize(d6[..., 0:3], p=2, dim=-1)
    z = F.normalize(torch.cross(x, d6[..., 3:6], dim=-1), p=2, dim=-1)
    y = torch.cross(z, x, dim=-1)
    return torch.stack((x, y, z), dim=-1)


def axangle2mat_ref(axis, angle):
    """transforms3d 0.4.1 axangles.axangle2mat (Rodrigues; normalises the axis)."""
    x, y, z = (float(a) for a in axis)
    n = math.sqrt(x * x + y * y + z * z)
    x, y, z = x / n, y / n, z / n
    c, s = math.cos(angle), math.sin(angle)
    C = 1 - c
    xs, ys, zs = x * s, y * s, z * s
    xC, yC, zC = x * C, y * C, z * C
    xyC, yzC, zxC = x * yC, y * zC, z * xC
    return np.array([[x * xC + c, xyC - zs, zxC + ys], [xyC + zs, y * yC + c, yzC - xs],
                     [zxC - ys, yzC + xs, z * zC + c]])


def pose_decode_ref(rot_m, pred_t, cam_K, bbox_center, resize_ratio, roi_wh, dataset="CAMERA+Real",
                    t_type="site"):
    """network/pose_utils/pose_from_pred_centroid_z.py:60-157 (pose_from_predictions_test, REL z,
    rot-mat branch) + network/pose_utils/utils.py:29-84 (allocentric_to_egocentric, mat->mat).
    Returns (rot_ego float32 (B,3,3), translation (B,3))."""
    cent = pred_t[:, :2] if t_type == "site" else pred_t[:, :2] * 0
    cx = (cent[:, 0] * roi_wh[:, 0] + bbox_center[:, 0]).unsqueeze(1)
    cy = (cent[:, 1] * roi_wh[:, 1] + bbox_center[:, 1]).unsqueeze(1)
    z = pred_t[:, 2:3] * resize_ratio.view(-1, 1)
    if dataset == "wild6d":
        z = z * cam_K[0, 0, 0] / 590
    trans = torch.cat([z * (cx - cam_K[:, 0:1, 2]) / cam_K[:, 0:1, 0],
                       z * (cy - cam_K[:, 1:2, 2]) / cam_K[:, 1:2, 1], z], dim=1)
    R = rot_m.detach().cpu().numpy()
    T = trans.detach().cpu().numpy()
    ego = np.zeros_like(R)
    cam_ray = np.asarray((0, 0, 1.0))
    for i in range(R.shape[0]):
        t = T[i]
        obj_ray = t.copy() / np.linalg.norm(t)
        angle = math.acos(cam_ray.dot(obj_ray))
        if angle > 0:
            ego[i] = np.dot(axangle2mat_ref(np.cross(cam_ray, obj_ray), angle), R[i])
        else:
            ego[i] = R[i]
    return torch.from_numpy(ego), trans


def posenet_forward_ref(P, data, cfg, return_intermediates=False):
    """network/PoseNet.py:173-231 (PoseNet.forward, do_loss=False).  ``data`` values are torch
    tensors with the eval-loader keys; returns the reference's dict (rot on CPU, as there)."""
    dt = next(iter(P.values())).dtype
    f = lambda k: data[k].to(dt)
    img = f("roi_img")
    mask_out = data["roi_mask"][..., :: cfg.img_size // cfg.out_res, :: cfg.img_size // cfg.out_res]  # Resize NEAREST :170,180
    feat = convnext_ref(P, img, cfg) if cfg.main_backbone == "convnext" else resnet34_ref(P, img)
    pred_size = size_head_ref(P, feat[0])
    nocs = xyz_head_ref(P, feat[0], "xyz_nocs_head.")
    nocs_feat = map_transformer_ref(P, nocs) if cfg.nocsmap_encoder == "att" else map_encoder_ref(P, nocs, cfg)
    red = F.conv2d(feat[0], P["feat_reducer.weight"], P["feat_reducer.bias"])
    ivfc = xyz_head_ref(P, torch.cat([red, nocs_feat], dim=1), "xyz_deform_head.")
    rot6d, pred_t = conv_pnp_ref(P, torch.cat([ivfc, f("roi_coord_2d")], dim=1))
    ms = f("mean_size")
    pred_size = pred_size + ms / ms.norm(dim=1).unsqueeze(-1)
    rot_m = rot6d_to_mat_ref(rot6d)
    rot, trans = pose_decode_ref(rot_m, pred_t, f("cam_K"), f("bbox_center"), f("resize_ratio"), f("roi_wh"),
                                 cfg.dataset, cfg.t_type)
    out = {"rot": rot, "trans": trans, "size": pred_size, "mask": mask_out, "nocs_coor": nocs, "ivfc_coor": ivfc}
    if return_intermediates:
        out.update(feat=feat[0], nocs_feat=nocs_feat, rot6d=rot6d, pred_t=pred_t, rot_allo=rot_m)
    return out


def load_params(state_dict_np, dtype=torch.float32):
    """numpy state dict (givepose_amd.synth.synth_state_dict) -> dict of torch tensors."""
    out = {}
    for k, v in state_dict_np.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out
