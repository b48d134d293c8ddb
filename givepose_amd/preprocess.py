"""Crop pre-processing on the device: frames + detections -> the `data` dict PoseNet.forward consumes.

Host side of `gp_crop_rois` (include/givepose_hip.h).  Mirrors evaluation/load_data_eval.py:253-288,318-333 of the
reference: per detection the square crop box (bbox centre, max side * DZI_PAD_SCALE clipped to the frame), the
affine map of tools/dataset_utils.py:116-157 (rot = 0) for the img_size and out_res crops, `roi_wh`
(tools/eval_utils.py:243-249), `resize_ratio`, `bbox_center`.  Only a few dozen scalars per detection are computed
here; the frames travel as uint8 (0.9 MB per 640x480 frame instead of 786 KB of fp32 per crop) and every pixel of
roi_img / roi_mask / roi_coord_2d is produced by the HIP kernel straight into the model's static input buffers.
"""
import numpy as np
import torch

from . import _lib

IMG_MEAN = (0.485, 0.456, 0.406)     # evaluation/load_data_eval.py:161-162
IMG_STD = (0.229, 0.224, 0.225)


def _affine_dst_from_src(cx, cy, src_w, dst):
    """2x3 map through the three point pairs of get_affine_transform(rot=0); points rounded to fp32 as there."""
    f = np.float32
    s0 = np.array([f(cx), f(cy)], dtype=np.float64)
    s1 = np.array([f(cx + 0.0), f(cy + src_w * -0.5)], dtype=np.float64)
    d0 = np.array([f(dst * 0.5), f(dst * 0.5)], dtype=np.float64)
    d1 = np.array([f(f(dst * 0.5) + f(0)), f(f(dst * 0.5) + f(dst * -0.5))], dtype=np.float64)

    def third(a, b):
        d = (a - b).astype(np.float32)
        return np.array([f(f(b[0]) - d[1]), f(f(b[1]) + d[0])], dtype=np.float64)

    S = np.stack([s0, s1, third(s0, s1)])
    D = np.stack([d0, d1, third(d0, d1)])
    A = np.concatenate([S, np.ones((3, 1))], axis=1)
    return np.stack([np.linalg.solve(A, D[:, 0]), np.linalg.solve(A, D[:, 1])])


def _invert(M):
    """cv::warpAffine's in-place inverse of a 2x3 map (double)."""
    a, b, c, d, e, g = [float(v) for v in M.reshape(6)]
    det = a * e - b * d
    det = 1.0 / det if det != 0 else 0.0
    A11, A22 = e * det, a * det
    b, d = b * -det, d * -det
    return np.array([A11, b, -A11 * c - b * g, d, A22, -d * c - A22 * g], dtype=np.float64)


def _affine_dst_from_src_batch(cx, cy, src_w, dst):
    """_affine_dst_from_src for n detections at once (same roundings; numpy's batched solve runs the same LAPACK gesv per
    3x3 system, so the maps are bit for bit those of the per-detection form)."""
    f = np.float32
    n = len(cx)
    s0 = np.stack([cx.astype(f), cy.astype(f)], 1).astype(np.float64)
    s1 = np.stack([(cx + 0.0).astype(f), (cy + src_w * -0.5).astype(f)], 1).astype(np.float64)
    h = f(dst * 0.5)
    d0 = np.broadcast_to(np.array([h, h], dtype=np.float64), (n, 2))
    d1 = np.broadcast_to(np.array([f(h + f(0)), f(h + f(dst * -0.5))], dtype=np.float64), (n, 2))

    def third(a, b):
        d = (a - b).astype(np.float32)
        return np.stack([(b[:, 0].astype(f) - d[:, 1]).astype(f), (b[:, 1].astype(f) + d[:, 0]).astype(f)], 1).astype(np.float64)

    S = np.stack([s0, s1, third(s0, s1)], 1)                       # (n, 3, 2)
    D = np.stack([d0, d1, third(d0, d1)], 1)
    A = np.concatenate([S, np.ones((n, 3, 1))], axis=2)            # (n, 3, 3)
    # one right-hand side per solve, as the per-detection form does (gesv with two right-hand sides at once orders its
    # eliminations differently: 1e-14 off, enough to move a fixed-point rounding of the warp)
    r0 = np.linalg.solve(A, D[:, :, 0:1])[:, :, 0]
    r1 = np.linalg.solve(A, D[:, :, 1:2])[:, :, 0]
    return np.stack([r0, r1], 1)                                   # (n, 2, 3)


def _invert_batch(M):
    """cv::warpAffine's in-place inverse of n 2x3 maps (double), element for element as _invert."""
    a, b, c, d, e, g = (M[:, 0, 0], M[:, 0, 1], M[:, 0, 2], M[:, 1, 0], M[:, 1, 1], M[:, 1, 2])
    det = a * e - b * d
    with np.errstate(divide="ignore"):
        det = np.where(det != 0, 1.0 / det, 0.0)
    A11, A22 = e * det, a * det
    b, d = b * -det, d * -det
    return np.stack([A11, b, -A11 * c - b * g, d, A22, -d * c - A22 * g], 1).astype(np.float64)


def crop_params(bboxes, im_H, im_W, img_size=256, out_res=64, pad_scale=1.5):
    """bboxes (n,4) as (y1,x1,y2,x2) -> dict of per-detection host arrays (float64 inverse maps + the model scalars);
    vectorised over the detections (a 64-detection batch costs ~0.2 ms of host time instead of ~6)."""
    bboxes = np.asarray(bboxes, dtype=np.float64).reshape(-1, 4)
    y1, x1, y2, x2 = bboxes[:, 0], bboxes[:, 1], bboxes[:, 2], bboxes[:, 3]
    ext = np.maximum(y2 - y1, x2 - x1)
    if not np.all(np.isfinite(bboxes)) or np.any(ext <= 0):      # a zero-extent box has no crop (scale 0: singular map, inf ratio);
        bad = np.nonzero(~(np.isfinite(bboxes).all(1) & (ext > 0)))[0]   # name it instead of failing the whole batch in the solve
        raise ValueError(f"crop_params: degenerate detection box(es) at index {bad.tolist()}: (y1,x1,y2,x2) = {bboxes[bad].tolist()}")
    cx, cy = 0.5 * (x1 + x2), 0.5 * (y1 + y2)
    scale = np.minimum(ext * pad_scale, max(im_H, im_W)) * 1.0
    inv_img = _invert_batch(_affine_dst_from_src_batch(cx, cy, scale, float(img_size)))
    inv_out = _invert_batch(_affine_dst_from_src_batch(cx, cy, scale, float(out_res)))
    wh = np.stack([np.minimum(im_W, x2) - np.maximum(0, x1), np.minimum(im_H, y2) - np.maximum(0, y1)], 1).astype(np.float32)
    ctr = np.stack([cx, cy], 1).astype(np.float32)
    ratio = (out_res / scale).astype(np.float32)
    return {"inv_img": inv_img, "inv_out": inv_out, "roi_wh": wh, "bbox_center": ctr, "resize_ratio": ratio}


def crop_params_loop(bboxes, im_H, im_W, img_size=256, out_res=64, pad_scale=1.5):
    """The per-detection form crop_params was vectorised from (kept as its cross-check, tests/test_preprocess.py)."""
    bboxes = np.asarray(bboxes, dtype=np.float64).reshape(-1, 4)
    n = len(bboxes)
    inv_img, inv_out = np.zeros((n, 6)), np.zeros((n, 6))
    wh, ctr, ratio = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32), np.zeros(n, np.float32)
    for j, (y1, x1, y2, x2) in enumerate(bboxes):
        cx, cy = 0.5 * (x1 + x2), 0.5 * (y1 + y2)
        scale = min(max(y2 - y1, x2 - x1) * pad_scale, max(im_H, im_W)) * 1.0
        inv_img[j] = _invert(_affine_dst_from_src(cx, cy, scale, float(img_size)))
        inv_out[j] = _invert(_affine_dst_from_src(cx, cy, scale, float(out_res)))
        wh[j] = (min(im_W, x2) - max(0, x1), min(im_H, y2) - max(0, y1))
        ctr[j] = (cx, cy)
        ratio[j] = out_res / scale
    return {"inv_img": inv_img, "inv_out": inv_out, "roi_wh": wh, "bbox_center": ctr, "resize_ratio": ratio}


def luts(im_H, im_W, mean=IMG_MEAN, std=IMG_STD):
    v = np.arange(256, dtype=np.float64)
    img = np.stack([((v / 255.0 - mean[c]) / std[c]) for c in range(3)]).astype(np.float32)
    x = np.linspace(0, im_W - 1, im_W, dtype=np.float32)
    y = np.linspace(0, im_H - 1, im_H, dtype=np.float32)
    x = ((x - np.float32((im_W - 1) / 2)) / np.float32((im_W - 1) / 2)).astype(np.float32)
    y = ((y - np.float32((im_H - 1) / 2)) / np.float32((im_H - 1) / 2)).astype(np.float32)
    return img, x, y


class RoiCropper:
    """Device-side replacement of the reference's per-detection cv2 crops.

        cropper = RoiCropper(480, 640, device)
        data = cropper(frames_u8, masks_u8, frame_idx, mask_idx, bboxes, out=static_inputs)   # -> data dict (device)
    """

    def __init__(self, im_H, im_W, device, img_size=256, out_res=64, pad_scale=1.5):
        self.H, self.W, self.S, self.R, self.pad = im_H, im_W, img_size, out_res, pad_scale
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("RoiCropper runs on a HIP device only (no CPU fallback)")
        if self.device.index is None:            # "cuda" -> "cuda:<current>": tensors report an indexed device
            self.device = torch.device("cuda", torch.cuda.current_device())
        il, xl, yl = luts(im_H, im_W)
        self.img_lut = torch.from_numpy(il).to(self.device)
        self.xlut = torch.from_numpy(xl).to(self.device)
        self.ylut = torch.from_numpy(yl).to(self.device)

    def __call__(self, frames, masks, frame_idx, mask_idx, bboxes, out=None):
        """frames (F,H,W,3) uint8 and masks (NM,H,W) uint8 on the device (or host: copied); frame_idx / mask_idx /
        bboxes: host sequences of length B.  Writes roi_img / roi_mask / roi_coord_2d / roi_wh / bbox_center /
        resize_ratio into `out` (e.g. PoseNet.static_inputs) or fresh tensors."""
        dev = self.device
        frames = torch.as_tensor(frames).to(dev, torch.uint8).contiguous()
        masks = torch.as_tensor(masks).to(dev, torch.uint8).contiguous()
        F, NM = frames.shape[0], masks.shape[0]
        if tuple(frames.shape[1:]) != (self.H, self.W, 3) or tuple(masks.shape[1:]) != (self.H, self.W):
            raise ValueError(f"frames {tuple(frames.shape)} / masks {tuple(masks.shape)} do not match ({self.H},{self.W})")
        fi = np.asarray(frame_idx, dtype=np.int32).reshape(-1)
        mi = np.asarray(mask_idx, dtype=np.int32).reshape(-1)
        B = len(fi)
        if len(mi) != B or len(bboxes) != B:
            raise ValueError("frame_idx, mask_idx and bboxes must have one entry per detection")
        if B == 0 or fi.min() < 0 or fi.max() >= F or mi.min() < 0 or mi.max() >= NM:
            raise ValueError("frame_idx / mask_idx out of range")      # the kernel trusts them
        P = crop_params(bboxes, self.H, self.W, self.S, self.R, self.pad)
        if out is None:
            out = {}
        def buf(name, shape):
            t = out.get(name)
            if t is None:
                t = out[name] = torch.empty(shape, device=dev, dtype=torch.float32)
            if tuple(t.shape) != tuple(shape) or t.dtype != torch.float32 or not t.is_contiguous() or t.device != dev:
                raise ValueError(f"out[{name!r}] must be a contiguous fp32 {tuple(shape)} tensor on {dev}")
            return t
        roi_img, roi_mask = buf("roi_img", (B, 3, self.S, self.S)), buf("roi_mask", (B, 1, self.S, self.S))
        roi_coord = buf("roi_coord_2d", (B, 2, self.R, self.R))
        host = torch.from_numpy(np.concatenate([P["inv_img"].reshape(-1), P["inv_out"].reshape(-1)])).pin_memory().to(dev, non_blocking=True)
        idx = torch.from_numpy(np.concatenate([fi, mi])).pin_memory().to(dev, non_blocking=True)
        L = _lib.load()
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(L.gp_crop_rois(frames.data_ptr(), masks.data_ptr(), idx.data_ptr(), idx.data_ptr() + 4 * B,
                                  host.data_ptr(), host.data_ptr() + 8 * 6 * B, self.img_lut.data_ptr(), self.xlut.data_ptr(),
                                  self.ylut.data_ptr(), roi_img.data_ptr(), roi_mask.data_ptr(), roi_coord.data_ptr(),
                                  B, F, NM, self.H, self.W, self.S, self.R, stream), "gp_crop_rois")
        for k in ("roi_wh", "bbox_center", "resize_ratio"):
            buf(k, P[k].shape).copy_(torch.from_numpy(P[k]).pin_memory(), non_blocking=True)
        return out
