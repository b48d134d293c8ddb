"""Thin tensor->pointer marshalling over the C ABI (include/givepose_hip.h).

Every function launches hand-written HIP kernels from libgivepose_hip.so on torch's current HIP
stream, writes into caller-provided (or freshly torch.empty'd) buffers and never falls back to a
PyTorch implementation: a missing library raises at import of the first op.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import (ACT_GELU, ACT_LRELU, ACT_NONE, ACT_RELU, EPI_GELU, EPI_LNFOLD_GELU, EPI_LRELU, EPI_NONE, EPI_RELU, EPI_RES_RELU,
                   EPI_SCALE_RES, GP_F16, GP_F32, GP_F64, GemmDesc, check)

__all__ = ["dtype_code", "gemm", "conv2d_nhwc", "dcnv3_forward", "dcnv3_backward", "dcnv3_forward_into", "convnext_stem", "dwconv_ln",
           "layernorm", "groupnorm", "upsample_bilinear2x", "deconv_col2im", "xyz_out_layer", "pointwise_k3",
           "pnp_conv1", "xyz_conv3x3_s2", "size_head", "pose_tail", "mask_resize_nearest"]


def _L():
    return _lib.load()


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dtype_code(dt):
    if dt == torch.float16:
        return GP_F16
    if dt == torch.float32:
        return GP_F32
    raise TypeError(f"unsupported dtype {dt} (float16 / float32)")


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _chk(t, name, dtype=None):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor")  # dcnv3_cuda.cu:32-34
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    return t


def _contig(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} tensor has to be contiguous")  # dcnv3_cuda.cu:29-31
    return t


# ----------------------------------------------------------------------------------- GEMM / conv
_WS = {}
_WS_RETIRED = []      # outgrown workspaces: a captured hipGraph may have their address baked in, so they are never freed


def _workspace(device, nfloats):
    """Split-K partial-sum workspace of the CURRENT stream: launches on different streams must not share one.  A
    workspace is only ever replaced by a larger one, and the old tensor is kept alive (an earlier hipGraph of that stream
    still points at it).  No allocation may happen while a hipGraph is being captured (the capture is raw HIP, which
    PyTorch's allocator does not know about): the eager warm-up pass in front of every capture sizes the workspace."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nfloats:
        if _lib.capturing():
            raise RuntimeError("split-K workspace allocation during hipGraph capture (run the same shapes eagerly once first)")
        if ws is not None:
            _WS_RETIRED.append(ws)
        ws = torch.empty(max(nfloats, 1 << 24), dtype=torch.float32, device=device)
        _WS[key] = ws
    return ws


# ----------------------------------------------------------------------------------- split-operand mode
SPLIT_SHIFT = 11        # lo' = fp16((v - fp16(v)) * 2^11): the low plane of a value sits in fp16's normal range whenever the high one does


class SplitW:
    """GEMM weights as the two fp16 planes of the split-operand mode (include/givepose_hip.h, gp_gemm_desc.split_shift):
    ``planes`` (2, N, K) fp16 = [hi, lo'], made once on the host from the fp32 checkpoint tensor."""

    def __init__(self, planes, shift=SPLIT_SHIFT):
        self.planes, self.shift = planes, shift
        self.shape = tuple(planes.shape[1:])
        self.dtype, self.device = torch.float32, planes.device     # what the layer computes in / where

    def numel(self):
        return self.planes.numel()

    def element_size(self):
        return 2

    def data_ptr(self):
        return self.planes.data_ptr()


def split_weights(w, device, shift=SPLIT_SHIFT):
    """fp32 (N, K) host tensor -> SplitW on `device` (host arithmetic: exact subtraction, power-of-two scale)."""
    w = w.detach().to(device="cpu", dtype=torch.float32).contiguous()
    hi = w.half()
    lo = ((w - hi.float()) * float(2 ** shift)).half()
    return SplitW(torch.stack([hi, lo], 0).contiguous().to(device), shift)


_SPLIT_WS = {}
_SPLIT_WS_RETIRED = []


def _split_scratch(device, nhalfs):
    """Per-stream scratch for the X planes of a split-operand GEMM (same lifetime rules as the split-K workspace)."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _SPLIT_WS.get(key)
    if ws is None or ws.numel() < nhalfs:
        if _lib.capturing():
            raise RuntimeError("split-plane scratch allocation during hipGraph capture (run the same shapes eagerly once first)")
        if ws is not None:
            _SPLIT_WS_RETIRED.append(ws)
        ws = torch.empty(max(nhalfs, 1 << 22), dtype=torch.float16, device=device)
        _SPLIT_WS[key] = ws
    return ws


def split_planes(x, rows, cols, ldx, out=None, shift=SPLIT_SHIFT):
    """fp32 (rows, cols) with row stride ldx -> fp16 planes [hi | lo'] (2, rows, cols) in `out` (default: the stream's scratch)."""
    _chk(x, "x", torch.float32)
    n = rows * cols
    if out is None:
        out = _split_scratch(x.device, 2 * n)
    check(_L().gp_split_planes(_ptr(x), _ptr(out), rows, cols, ldx, n, shift, _stream()), "gp_split_planes")
    return out


# Automatic split-K for skinny GEMMs (the PnP fc layers, feat_reducer): 128x128 LDS-DMA tiles + a reduce kernel.
# (Round 1 ran split-K on a register-staged kernel whose MFMA loop corrupted packed-fp32 results of OTHER kernels' waves
# on the same SIMD when batches overlapped -- that kernel is gone)
AUTO_SPLITK = True
# set beside it: tells gp_gemm that other launches run next to this one (tile chosen per FLOP, not to fill the chip)
CO_SCHEDULED = False


def auto_splitk(M, N, K, esz, n_cu=256, plain=True):
    """Split only long-K GEMMs with a handful of tiles (the PnP fc1: 16 tiles, 128 K steps -> 16 ranges, 85 -> 22 us);
    measured on MI355X (scripts/splitk_bench.py): with >= 32 tiles or < 32 K steps the reduce kernel costs more than the
    split saves (feat_reducer 14.6 -> 22.7 us, fc2 16.7 -> 16.3 us)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    nkt = K // (128 // esz)
    # round 6 (scripts/stage3_splitk.py, profiles/r06_stage3_splitk.txt): ConvNeXt stage-3 fc2 (N = 1024, K = 4096) at 16 .. 32 crops is 64 .. 128 tiles of 128 x 128
    # with a 64-step K loop on a quarter to a half of the chip: four K ranges + the reduce kernel 43.6 -> 28.1 / 43.9 -> 31.4 / 43.9 -> 33.9 us (from 192 tiles: nothing)
    # (plain GEMMs only: measured on that shape; the ResNet-34 variant's 3x3 x 512 convs have K = 4 608 and were not)
    if plain and esz == 2 and K >= 4096 and 32 < tiles <= 128:
        return 4
    if tiles >= 32 or nkt < 32:
        return 1
    return max(1, min(nkt // 4, (n_cu + tiles - 1) // tiles, 32))


def gemm(x, w, out, bias=None, epilogue=EPI_NONE, gamma=None, residual=None, M=None, K=None, ldx=None, ldc=None,
         ldres=None, conv=None, splitk=None, variant=0, gn=None, ln=None, prefetch=None, _stamps=None, x_planes=False,
         out_planes=False, out16=None):
    """out[m][n] = epi(sum_k x[m][k] w[n][k] + bias[n]).  ``x``/``w`` share dtype (f16|f32); ``out`` is that
    dtype or float32.  conv = dict(B,H,W,Cin,KH,KW,stride,pad) switches X to channels-last implicit GEMM.
    prefetch = a tensor (the weights of the NEXT launch on this stream) to be pulled towards the caches meanwhile: a hint.
    w a SplitW: split-operand mode (fp32 x / out, fp16 hi + lo' operand planes).  out_planes: `out` (fp32 (M, N) storage) receives
    the result as the two fp16 planes [hi (M, N) | lo' (M, N)] instead of fp32 values; x_planes: `x` is such a container.
    fp16 x / w with a float32 `residual` and float32 `out`: the fp32 residual stream of the fp16 mode (gp_gemm_desc.residual_f32);
    out16: a float16 tensor that receives the same output values rounded (gp_gemm_desc.c16), for the consumers that read fp16."""
    split = isinstance(w, SplitW)
    dt = x.dtype
    if split:
        # fp32 activations in, fp32 out; the MFMA operands are fp16 hi / lo' planes (X split here, W by the host)
        _chk(x, "x", torch.float32), _chk(out, "out", torch.float32)
        if residual is not None:
            _chk(residual, "residual", torch.float32)
        code = GP_F16
    else:
        code = dtype_code(dt)
        _chk(x, "x"), _chk(w, "w", dt), _chk(out, "out")
    N = w.shape[0]
    Kw = w.shape[1]
    d = GemmDesc()
    if conv is not None:
        B, H, W_, Cin, KH, KW, stride, pad = (conv[k] for k in ("B", "H", "W", "Cin", "KH", "KW", "stride", "pad"))
        Ho = (H + 2 * pad - KH) // stride + 1
        Wo = (W_ + 2 * pad - KW) // stride + 1
        M = B * Ho * Wo
        K = KH * KW * Cin
        d.B, d.H, d.Win, d.Cin, d.KH, d.KW, d.stride, d.pad, d.Ho, d.Wo = B, H, W_, Cin, KH, KW, stride, pad, Ho, Wo
        ldx = 0
    else:
        M = x.shape[0] if M is None else M
        K = Kw if K is None else K
        ldx = x.stride(0) if ldx is None else ldx
    assert K == Kw, (K, Kw)
    xptr = x.data_ptr()
    if split:
        if conv is not None:
            if not x.is_contiguous():
                raise RuntimeError("gemm(split, conv): x must be contiguous")
            rows, cols, ld = x.numel() // conv["Cin"], conv["Cin"], conv["Cin"]
        else:
            rows, cols, ld = M, K, ldx
            ldx = K                      # the planes are dense
        if x_planes:     # written by the previous split-operand GEMM (out_planes): [hi (rows, cols) | lo' (rows, cols)] in x's storage
            if ld != cols or not x.is_contiguous():
                raise RuntimeError("gemm(x_planes): dense operand only")
        else:
            xptr = split_planes(x, rows, cols, ld, shift=w.shift).data_ptr()
        d.split_shift, d.x_plane_stride, d.w_plane_stride = w.shift, rows * cols, N * K
        out_f32 = 1
        if out_planes:
            if not out.is_contiguous() or out.shape[-1] != N or (ldc is not None and ldc != N) or splitk not in (None, 1):
                raise RuntimeError("gemm(out_planes): dense (M, N) fp32 storage, no split-K")
            d.out_planes, d.c_plane_stride = 1, M * N
            splitk = 1
    else:
        out_f32 = 1 if (out.dtype == torch.float32 and dt != torch.float32) else 0
        if out.dtype not in (dt, torch.float32):
            raise TypeError("gemm: out dtype must equal x dtype or float32")
    ldc = out.stride(0) if ldc is None else ldc
    esz = 2 if code == GP_F16 else 4
    if (x_planes or out_planes) and not split:
        raise RuntimeError("gemm: x_planes / out_planes belong to the split-operand mode")
    if splitk is None:
        splitk = auto_splitk(M, N, K, esz, plain=conv is None) if (AUTO_SPLITK and variant in (0, 4) and gn is None and ln is None) else 1
    d.X, d.W, d.C = xptr, w.data_ptr(), out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.gamma = gamma.data_ptr() if gamma is not None else None
    d.residual = residual.data_ptr() if residual is not None else None
    if splitk > 1:
        d.workspace = _workspace(x.device, splitk * M * N).data_ptr()
    elif _stamps is not None:      # in-kernel time stamps of variant 1616 (scripts/wreg_stamps.py)
        d.workspace = _stamps.data_ptr()
    d.M, d.N, d.K, d.ldx, d.ldc = M, N, K, ldx, ldc
    d.ldres = (residual.stride(0) if ldres is None else ldres) if residual is not None else 0
    if not split and residual is not None and residual.dtype == torch.float32 and dt == torch.float16:
        if out.dtype != torch.float32:
            raise TypeError("gemm: an fp32 residual goes with an fp32 output")
        d.residual_f32 = 1
        splitk = 1
    if out16 is not None:
        if out16.dtype != torch.float16 or out.dtype != torch.float32 or dt != torch.float16 or split:
            raise TypeError("gemm(out16): fp16 operands, float32 out, float16 out16")
        d.c16, d.ldc16 = out16.data_ptr(), out16.stride(0)
        splitk = 1
    d.epilogue, d.out_f32, d.splitk, d.dtype, d.variant = epilogue, out_f32, splitk, code, variant
    d.co_scheduled = 1 if CO_SCHEDULED else 0
    if prefetch is not None:
        d.prefetch, d.prefetch_bytes = prefetch.data_ptr(), prefetch.numel() * prefetch.element_size()
    if ln is not None:       # (row moments (M,2,nslab), column sums of w, nslab, eps): EPI_LNFOLD_GELU
        d.ln_stats, d.ln_colsum, d.ln_nslab, d.ln_eps = ln[0].data_ptr(), ln[1].data_ptr(), ln[2], ln[3]
    if gn is not None:       # (partial buffer, groups, pixels per image[, rows per statistics chunk]): fused GroupNorm statistics of the output
        d.gn_partial, d.gn_groups, d.gn_hw = gn[0].data_ptr(), gn[1], gn[2]
        d.gn_rows = gn[3] if len(gn) > 3 else 0
    check(_L().gp_gemm(ctypes.byref(d), _stream()), "gp_gemm")
    return out


def conv2d_nhwc(x, w_packed, KH, KW, stride, pad, out=None, bias=None, epilogue=EPI_NONE, variant=0, gn=None,
                residual=None, prefetch=None, x_planes=False, out16=None):
    """Channels-last convolution: x (B,H,W,Cin), w_packed (Cout, KH*KW*Cin) with K = (kh*KW+kw)*Cin+ci."""
    B, H, W_, Cin = x.shape
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W_ + 2 * pad - KW) // stride + 1
    if out is None:
        out = torch.empty(B, Ho, Wo, w_packed.shape[0], dtype=x.dtype, device=x.device)
    gemm(x, w_packed, out.view(B * Ho * Wo, -1), bias=bias, epilogue=epilogue,
         residual=None if residual is None else residual.view(B * Ho * Wo, -1),
         conv=dict(B=B, H=H, W=W_, Cin=Cin, KH=KH, KW=KW, stride=stride, pad=pad), variant=variant, gn=gn, prefetch=prefetch,
         x_planes=x_planes, out16=None if out16 is None else out16.view(B * Ho * Wo, -1))
    return out


# ----------------------------------------------------------------------------------- DCNv3
def dcnv3_forward_into(inp, offset, mask, out, K, stride, pad, dil, G, D, offset_scale, im2col_step=256,
                       remove_center=0, off_ld=None, mask_ld=None, mask_is_logits=False):
    _contig(_chk(inp, "input"), "input"), _chk(offset, "offset"), _chk(mask, "mask"), _contig(_chk(out, "out"), "out")
    N, H, W_, C = inp.shape
    P = K * K - int(remove_center)
    if C != G * D:
        raise RuntimeError(f"Input channels and group times group channels wont match: ({C} vs {G * D}).")
    if offset.dtype != mask.dtype:
        raise TypeError("offset and mask must share a dtype")
    check(_L().gp_dcnv3_forward(_ptr(inp), _ptr(offset), _ptr(mask), _ptr(out), N, H, W_, G, D, K, stride, pad, dil,
                                float(offset_scale), int(remove_center), int(im2col_step),
                                G * P * 2 if off_ld is None else off_ld, G * P if mask_ld is None else mask_ld,
                                1 if mask_is_logits else 0, dtype_code(inp.dtype), dtype_code(offset.dtype), _stream()),
          "gp_dcnv3_forward")
    return out


OUT_PLANES = 0x100     # include/givepose_hip.h GP_OUT_PLANES (fp32 producers writing split-operand planes; shift GP_SPLIT_SHIFT = 11)


def _planes_code(t, out_planes, out=None):
    """dtype code of an fp32 producer kernel, with GP_OUT_PLANES when its (dense) output is to be the hi / lo' planes a
    split-operand gemm(..., x_planes=True) reads."""
    code = dtype_code(t.dtype)
    if out_planes:
        if t.dtype != torch.float32 or SPLIT_SHIFT != 11:
            raise TypeError("out_planes: float32 storage (split-operand mode) only")
        if out is not None and out.data_ptr() == t.data_ptr():
            raise RuntimeError("out_planes: the output must not alias the input")
        code |= OUT_PLANES
    return code


def _any_dtype(t):
    if t.dtype == torch.float64:
        return GP_F64
    return dtype_code(t.dtype)


def _dcn_out_hw(H, W_, kh, kw, sh, sw, ph, pw, dh, dw):
    return (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1, (W_ + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1


def dcnv3_forward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                  group, group_channels, offset_scale, im2col_step, remove_center=0):
    """Drop-in for the reference pybind op ``DCNv3.dcnv3_forward`` (network/ops_dcnv3/src/vision.cpp:15,
    dcnv3.h:20-38; called from functions/dcnv3_func.py:53): same argument list, channels-last tensors of
    float64 / float32 / float16, returns a freshly allocated (N,Ho,Wo,G*D) tensor.  offset/mask are consumed as flat
    buffers exactly like the CUDA kernel.  The PoseNet geometry (square kernel, group_channels % 4 == 0, fp16 / fp32) runs
    on the wave kernels of csrc/dcnv3.hip, everything else on the generic kernel of csrc/dcnv3_any.hip."""
    _contig(_chk(input, "input"), "input"), _contig(_chk(offset, "offset"), "offset"), _contig(_chk(mask, "mask"), "mask")
    N, H, W_, C = input.shape
    if C != group * group_channels:
        raise RuntimeError(f"Input channels and group times group channels wont match: ({C} vs {group * group_channels}).")
    Ho, Wo = _dcn_out_hw(H, W_, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w)
    need = N * Ho * Wo * group * (kernel_h * kernel_w - int(remove_center))
    if offset.numel() < need * 2 or mask.numel() < need:
        raise RuntimeError("offset/mask buffers smaller than the kernel consumes")
    out = torch.empty(N, Ho, Wo, C, dtype=input.dtype, device=input.device)
    square = kernel_h == kernel_w and stride_h == stride_w and pad_h == pad_w and dilation_h == dilation_w
    if square and group_channels % 4 == 0 and input.dtype in (torch.float16, torch.float32):
        return dcnv3_forward_into(input, offset, mask, out, kernel_h, stride_h, pad_h, dilation_h, group, group_channels,
                                  offset_scale, im2col_step, remove_center)
    if not (offset.dtype == mask.dtype == input.dtype):
        raise TypeError("input, offset and mask must share a dtype")
    check(_L().gp_dcnv3_forward_any(_ptr(input), _ptr(offset), _ptr(mask), _ptr(out), N, H, W_, group, group_channels,
                                    kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                                    float(offset_scale), int(remove_center), int(im2col_step), _any_dtype(input), _stream()),
          "gp_dcnv3_forward_any")
    return out


def dcnv3_backward(input, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w,
                   group, group_channels, offset_scale, grad_output, im2col_step, remove_center=0):
    """Drop-in for ``DCNv3.dcnv3_backward`` (network/ops_dcnv3/src/dcnv3.h:40-59; called from functions/dcnv3_func.py:79-84):
    returns [grad_input, grad_offset, grad_mask] shaped like input / offset / mask, in the input dtype (computed in
    opmath precision and, for float16, rounded at the end: dcnv3_cuda.cu:123-126, 167-173)."""
    for t, n in ((input, "input"), (offset, "offset"), (mask, "mask"), (grad_output, "grad_output")):
        _contig(_chk(t, n), n)
    if not (offset.dtype == mask.dtype == grad_output.dtype == input.dtype):
        raise TypeError("input, offset, mask and grad_output must share a dtype")
    N, H, W_, C = input.shape
    if C != group * group_channels:
        raise RuntimeError(f"Input channels and group times group channels wont match: ({C} vs {group * group_channels}).")
    Ho, Wo = _dcn_out_hw(H, W_, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w)
    if tuple(grad_output.shape) != (N, Ho, Wo, C):
        raise RuntimeError(f"grad_output shape {tuple(grad_output.shape)} != {(N, Ho, Wo, C)}")
    acc = torch.float64 if input.dtype == torch.float64 else torch.float32
    gi = torch.empty(input.shape, dtype=acc, device=input.device)
    go = torch.empty(offset.shape, dtype=acc, device=input.device)
    gm = torch.empty(mask.shape, dtype=acc, device=input.device)
    check(_L().gp_dcnv3_backward(_ptr(input), _ptr(offset), _ptr(mask), _ptr(grad_output), _ptr(gi), _ptr(go), _ptr(gm),
                                 offset.numel(), mask.numel(), N, H, W_, group, group_channels, kernel_h, kernel_w, stride_h,
                                 stride_w, pad_h, pad_w, dilation_h, dilation_w, float(offset_scale), int(remove_center),
                                 int(im2col_step), _any_dtype(input), _stream()), "gp_dcnv3_backward")
    if input.dtype == torch.float16:
        return [gi.half(), go.half(), gm.half()]
    return [gi, go, gm]


# ----------------------------------------------------------------------------------- norms & small ops
def convnext_stem(img, w, b, ln_w, ln_b, out, eps=1e-6):
    B, _, H, W_ = img.shape
    _contig(_chk(img, "img", torch.float32), "img")
    check(_L().gp_convnext_stem(_ptr(img), _ptr(w), _ptr(b), _ptr(ln_w), _ptr(ln_b), _ptr(out), B, H, W_, w.shape[1],
                                eps, dtype_code(out.dtype), _stream()), "gp_convnext_stem")
    return out


MLP_S32 = 0x400        # include/givepose_hip.h GP_MLP_S32


def convnext_mlp_pack_w2(w2, s32=False):
    """fc2.weight (C, 4C) fp16 -> the column order gp_convnext_mlp reads (include/givepose_hip.h); s32: the order of its 32x32x16-MFMA form."""
    out = torch.empty_like(w2)
    fn = _L().gp_convnext_mlp_pack_w2_s32 if s32 else _L().gp_convnext_mlp_pack_w2
    check(fn(_ptr(_contig(w2, "w2")), _ptr(out), w2.shape[0], _stream()), "gp_convnext_mlp_pack_w2")
    return out


def convnext_mlp(x, w1, b1, w2p, b2, gamma, residual, out, s32=False):
    """out = residual + gamma * fc2(GELU(fc1(x))) in one launch (fp16, C in {128, 256}); out may alias residual.  s32: w2p was packed with s32=True."""
    M, C = x.shape
    check(_L().gp_convnext_mlp(_ptr(_contig(x, "x")), _ptr(w1), _ptr(b1), _ptr(w2p), _ptr(b2), _ptr(gamma), _ptr(residual),
                               _ptr(out), M, C, dtype_code(x.dtype) | (MLP_S32 if s32 else 0), _stream()), "gp_convnext_mlp")
    return out


def dwconv_ln(x, wt, bias, ln_w, ln_b, out, KS, eps=1e-6, act=ACT_NONE, n_pixels=None, out_planes=False):
    B, H, W_, C = x.shape
    n = B * H * W_ if n_pixels is None else n_pixels
    if out_planes and n != B * H * W_:
        raise RuntimeError("dwconv_ln(out_planes): whole tensor only")
    check(_L().gp_dwconv_ln(_ptr(_contig(x, "x")), _ptr(wt), _ptr(bias), _ptr(ln_w), _ptr(ln_b), _ptr(out), B, H, W_, C, KS,
                            eps, act, n, _planes_code(x, out_planes, out), _stream()), "gp_dwconv_ln")
    return out


def dwconv_ln_groups(x, wt, bias, ln_w, ln_b, out, KS, crop_group_start, eps=1e-6, act=ACT_NONE):
    """gp_dwconv_ln_groups: the quarter-size prefix rows of several batches in one launch; crop_group_start: int32 (B,) on the device."""
    B, H, W_, C = x.shape
    if crop_group_start.dtype != torch.int32 or crop_group_start.numel() < B or not crop_group_start.is_cuda:
        raise RuntimeError("dwconv_ln_groups: crop_group_start must be a device int32 tensor with one entry per crop")
    if out.numel() < B * H * W_ // 4 * C:
        raise RuntimeError("dwconv_ln_groups: output too small")
    check(_L().gp_dwconv_ln_groups(_ptr(_contig(x, "x")), _ptr(wt), _ptr(bias), _ptr(ln_w), _ptr(ln_b), _ptr(out), B, H, W_, C, KS, eps, act,
                                   _ptr(crop_group_start), dtype_code(x.dtype), _stream()), "gp_dwconv_ln_groups")
    return out


def dwconv7_raw_stats(x, wt, bias, out, stats):
    """Depth-wise 7x7 + bias only; per-pixel slab moments of the rounded output -> stats (pixels, 2, C/128) fp32."""
    B, H, W_, C = x.shape
    check(_L().gp_dwconv7_raw_stats(_ptr(_contig(x, "x")), _ptr(wt), _ptr(bias), _ptr(out), _ptr(stats), B, H, W_, C,
                                    dtype_code(x.dtype), _stream()), "gp_dwconv7_raw_stats")
    return out


IN_F32 = 0x200         # include/givepose_hip.h GP_IN_F32


def layernorm(x, w, b, out, eps=1e-6, ldy=0, out_planes=False):
    C = x.shape[-1]
    rows = x.numel() // C
    if x.dtype == torch.float32 and out.dtype == torch.float16:        # fp32 residual stream of the fp16 mode -> fp16 branch
        code = GP_F16 | IN_F32
    else:
        code = _planes_code(x, out_planes, out)
    check(_L().gp_layernorm(_ptr(_contig(x, "x")), _ptr(w), _ptr(b), _ptr(out), rows, C, eps, ldy, code, _stream()), "gp_layernorm")
    return out


def patchify_xyz(xyz4, out, B, R, P):
    check(_L().gp_patchify_xyz(_ptr(xyz4), _ptr(out), B, R, P, dtype_code(out.dtype), _stream()), "gp_patchify_xyz")
    return out


def attention64(qkv, out, B, heads):
    check(_L().gp_attention64(_ptr(_contig(qkv, "qkv")), _ptr(out), B, heads, dtype_code(qkv.dtype), _stream()), "gp_attention64")
    return out


def groupnorm_chunks(B, HW):
    return _L().gp_groupnorm_chunks(B, HW)


def gemm_gn_rows(M, N, K, HW):
    """Rows per fused-GroupNorm statistics chunk the library wants for an fp16 GEMM / conv of this shape (64, or 32 / 16 at few rows): the 4th entry
    of gemm(..., gn=(partial, G, HW, rows)) and the `rows` of the consumer (groupnorm(fused_stats=True), groupnorm_upsample2x, groupnorm_apply_xyz)."""
    return _L().gp_gemm_gn_rows(M, N, K, HW)


def groupnorm(x, w, b, out, G, act, partial, eps=1e-5, ldy=None, fused_stats=False, out_planes=False, rows=64):
    """x (B,HW,C) channels-last -> out rows of stride ldy (default C); in-place allowed.  fused_stats: ``partial``
    was already filled by the producing gemm(..., gn=(partial, G, HW[, rows])) in `rows`-row chunks."""
    B, HW, C = x.shape
    code = dtype_code(x.dtype)
    if not fused_stats:
        check(_L().gp_groupnorm_stats(_ptr(_contig(x, "x")), _ptr(partial), B, HW, C, G, code, _stream()), "gp_groupnorm_stats")
    check(_L().gp_groupnorm_apply(_ptr(x), _ptr(partial), _ptr(w), _ptr(b), _ptr(out), B, HW, C, G, eps, act,
                                  C if ldy is None else ldy, HW // rows if fused_stats else 0, _planes_code(x, out_planes, out), _stream()),
          "gp_groupnorm_apply")
    return out


def groupnorm_upsample2x(x, w, b, out, G, act, partial, eps=1e-5, rows=64):
    """GroupNorm apply (statistics already in ``partial`` in `rows`-row chunks, from the producing conv) + act + bilinear x2
    (align_corners) in one pass: x (B,H,W,C) fp16 -> out (B,2H,2W,C) fp16; bitwise groupnorm(fused_stats=True) + upsample_bilinear2x."""
    B, H, W_, C = x.shape
    check(_L().gp_groupnorm_upsample2x(_ptr(_contig(x, "x")), _ptr(partial), _ptr(w), _ptr(b), _ptr(out), B, H, W_, C, G, eps, act,
                                       H * W_ // rows, dtype_code(x.dtype), _stream()), "gp_groupnorm_upsample2x")
    return out


def groupnorm_apply_xyz(x, w, b, out_w, out_b, out_nchw, out_nhwc4, G, act, partial, eps=1e-5, rows=64, packed16=False):
    """packed16 (fp16, C = 256, GELU): affine + GELU on packed fp16 arithmetic (GP_ACT_PACKED16), the fp16 mode's form.
    GroupNorm apply (statistics already in ``partial`` in `rows`-row chunks) + act + 1x1 out layer, nothing else written."""
    B, HW, C = x.shape
    check(_L().gp_groupnorm_apply_xyz(_ptr(_contig(x, "x")), _ptr(partial), _ptr(w), _ptr(b), _ptr(out_w), _ptr(out_b),
                                      _ptr(out_nchw), _ptr(out_nhwc4), B, HW, C, G, eps, act | (0x100 if packed16 else 0), HW // rows, dtype_code(x.dtype),
                                      _stream()), "gp_groupnorm_apply_xyz")


def upsample_bilinear2x(x, out, out_planes=False):
    B, H, W_, C = x.shape
    check(_L().gp_upsample_bilinear2x(_ptr(_contig(x, "x")), _ptr(out), B, H, W_, C, _planes_code(x, out_planes, out), _stream()),
          "gp_upsample_bilinear2x")
    return out


def deconv_col2im(cols, out, B, H, W_, C):
    """cols fp32, or (fp16 mode, round 5) fp16 like `out`: GP_COLS_F16."""
    code = dtype_code(out.dtype) | (0x400 if cols.dtype == torch.float16 else 0)
    check(_L().gp_deconv_col2im(_ptr(cols), _ptr(out), B, H, W_, C, code, _stream()), "gp_deconv_col2im")
    return out


def xyz_out_layer(x, w, b, out_nchw, out_nhwc4):
    B, HW, C = x.shape
    check(_L().gp_xyz_out_layer(_ptr(_contig(x, "x")), _ptr(w), _ptr(b), _ptr(out_nchw), _ptr(out_nhwc4), B, HW, C,
                                dtype_code(x.dtype), _stream()), "gp_xyz_out_layer")


def pointwise_k3(xyz4, w, b, out):
    rows = xyz4.shape[0]
    check(_L().gp_pointwise_k3(_ptr(xyz4), _ptr(w), _ptr(b), _ptr(out), rows, w.shape[0], dtype_code(out.dtype), _stream()),
          "gp_pointwise_k3")
    return out


def pnp_conv1(xyz4, coord2d, w, out, B, R):
    check(_L().gp_pnp_conv1(_ptr(xyz4), _ptr(_contig(coord2d, "coord2d")), _ptr(w), _ptr(out), B, R, w.shape[1],
                            dtype_code(out.dtype), _stream()), "gp_pnp_conv1")
    return out


def xyz_conv3x3_s2(xyz4, w, out, B, R):
    check(_L().gp_xyz_conv3x3_s2(_ptr(xyz4), _ptr(w), _ptr(out), B, R, w.shape[1], dtype_code(out.dtype), _stream()),
          "gp_xyz_conv3x3_s2")
    return out


def size_head(feat, w1, b1, w2, b2, mean_size, out, scratch):
    B, HW, C = feat.shape
    if scratch.numel() < B * (w1.shape[0] + C):
        raise ValueError("size_head: scratch needs B * (F + C) floats")
    check(_L().gp_size_head(_ptr(_contig(feat, "feat")), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(mean_size), _ptr(out),
                            _ptr(scratch), B, HW, C, w1.shape[0], dtype_code(feat.dtype), _stream()), "gp_size_head")
    return out


def pose_tail(h, hz, ldh, W, cam_K, bbox_center, resize_ratio, roi_wh, wild6d, site, outs, B):
    check(_L().gp_pose_tail(_ptr(h), _ptr(hz), ldh, _ptr(W["fc_r.w"]), _ptr(W["fc_r.b"]), _ptr(W["fc_t.w"]), _ptr(W["fc_t.b"]),
                            _ptr(W["fc_z.w"]), _ptr(W["fc_z.b"]), _ptr(cam_K), _ptr(bbox_center), _ptr(resize_ratio),
                            _ptr(roi_wh), int(wild6d), int(site), _ptr(outs["rot6d"]), _ptr(outs["pred_t"]),
                            _ptr(outs["rot_allo"]), _ptr(outs["rot_ego"]), _ptr(outs["trans"]), B, _stream()), "gp_pose_tail")


def resnet_stem(img, w, b, out):
    B, _, H, W_ = img.shape
    check(_L().gp_resnet_stem(_ptr(_contig(_chk(img, "img", torch.float32), "img")), _ptr(w), _ptr(b), _ptr(out), B, H, W_,
                              dtype_code(out.dtype), _stream()), "gp_resnet_stem")
    return out


def maxpool3x3s2(x, out):
    B, H, W_, C = x.shape
    check(_L().gp_maxpool3x3s2(_ptr(_contig(x, "x")), _ptr(out), B, H, W_, C, dtype_code(x.dtype), _stream()), "gp_maxpool3x3s2")
    return out


def mask_resize_nearest(mask, out):
    B, _, S, _ = mask.shape
    R = out.shape[-1]
    check(_L().gp_mask_resize_nearest(_ptr(_contig(_chk(mask, "mask", torch.float32), "mask")), _ptr(out), B, S, R, _stream()),
          "gp_mask_resize_nearest")
    return out
