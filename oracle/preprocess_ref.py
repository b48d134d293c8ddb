"""TEST INFRASTRUCTURE ONLY (oracle): numpy restatement of the reference's crop pre-processing.

Follows evaluation/load_data_eval.py:253-288,318-333 (bbox -> center / scale, the four NEAREST crops, image
normalisation, roi_wh / resize_ratio / bbox_center) and tools/dataset_utils.py:8-30,101-157 (`get_2d_coord_np`,
`crop_resize_by_warp_affine`, `get_affine_transform`), tools/eval_utils.py:185-187,243-249 (`get_bbox_ori`,
`get_real_hw`).

The arithmetic of `cv2.getAffineTransform` / `cv2.warpAffine(..., flags=INTER_NEAREST)` lives in OpenCV
(opencv-python 4.8.0.76, GIVEPose_env.yml:250), which is NOT installed in this image: it is restated here from the
published algorithm (imgproc/src/imgwarp.cpp: double-precision inverse of the 2x3 matrix, fixed-point coordinates
with AB_BITS = 10, `X = (saturate_cast<int>((M1*y + M2) * 1024) + 512 + saturate_cast<int>(M0 * x * 1024)) >> 10`,
BORDER_CONSTANT 0).  **Parity unpinned** against cv2 itself; the reference holds no test or fixture for this step.
Only tests/ may import this module.
"""
import numpy as np

AB_BITS = 10
AB_SCALE = 1 << AB_BITS


def get_2d_coord_ref(width, height):
    """tools/dataset_utils.py:8-30 with norm=True, fmt CHW -> returned as (H, W, 2) like the caller's transpose."""
    x = np.linspace(0, width - 1, width, dtype=np.float32)
    y = np.linspace(0, height - 1, height, dtype=np.float32)
    x = ((x - np.float32((width - 1) / 2)) / np.float32((width - 1) / 2)).astype(np.float32)
    y = ((y - np.float32((height - 1) / 2)) / np.float32((height - 1) / 2)).astype(np.float32)
    xy = np.asarray(np.meshgrid(x, y))
    return xy.transpose(1, 2, 0)


def _third_point(a, b):
    d = a - b
    return b + np.array([-d[1], d[0]], dtype=np.float32)


def get_affine_transform_ref(center, scale, output_size):
    """tools/dataset_utils.py:116-157 with rot = 0, shift = 0, inv = False; cv2.getAffineTransform = the exact affine
    map through three point pairs, solved in float64 from float32 points."""
    center = np.asarray(center, dtype=np.float64)
    src_w = float(scale)
    dst_w, dst_h = float(output_size), float(output_size)
    src = np.zeros((3, 2), dtype=np.float32)
    dst = np.zeros((3, 2), dtype=np.float32)
    src_dir = np.array([0.0, src_w * -0.5])            # get_dir with rot_rad = 0: sn = 0, cs = 1
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src[0, :] = center
    src[1, :] = center + src_dir
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + dst_dir
    src[2, :] = _third_point(src[0, :], src[1, :])
    dst[2, :] = _third_point(dst[0, :], dst[1, :])
    A = np.zeros((6, 6), dtype=np.float64)
    b = np.zeros(6, dtype=np.float64)
    for i in range(3):                                   # cv::getAffineTransform's 6x6 system
        A[i, 0:3] = [src[i, 0], src[i, 1], 1.0]
        A[i + 3, 3:6] = [src[i, 0], src[i, 1], 1.0]
        b[i] = dst[i, 0]
        b[i + 3] = dst[i, 1]
    return np.linalg.solve(A, b).reshape(2, 3)


def invert_affine_ref(M):
    """The in-place inverse cv::warpAffine applies when WARP_INVERSE_MAP is not set."""
    M = np.array(M, dtype=np.float64).reshape(6).copy()
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M


def warp_affine_nearest_ref(img, M, out_size):
    """cv2.warpAffine(img, M, (out_size, out_size), flags=cv2.INTER_NEAREST), border constant 0."""
    iM = invert_affine_ref(M)
    xs = np.arange(out_size, dtype=np.float64)
    adelta = np.rint(iM[0] * xs * AB_SCALE).astype(np.int64)
    bdelta = np.rint(iM[3] * xs * AB_SCALE).astype(np.int64)
    X0 = np.rint((iM[1] * xs + iM[2]) * AB_SCALE).astype(np.int64) + AB_SCALE // 2      # indexed by y
    Y0 = np.rint((iM[4] * xs + iM[5]) * AB_SCALE).astype(np.int64) + AB_SCALE // 2
    X = (X0[:, None] + adelta[None, :]) >> AB_BITS
    Y = (Y0[:, None] + bdelta[None, :]) >> AB_BITS
    H, W = img.shape[:2]
    ok = (X >= 0) & (X < W) & (Y >= 0) & (Y < H)
    Xc, Yc = np.clip(X, 0, W - 1), np.clip(Y, 0, H - 1)
    out = img[Yc, Xc]
    out = np.where(ok.reshape(ok.shape + (1,) * (out.ndim - 2)), out, np.zeros((), dtype=img.dtype))
    return out


def crop_batch_ref(image, masks, bboxes, img_size=256, out_res=64, pad_scale=1.5,
                   mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """image (H,W,3) uint8, masks (H,W,n) bool/uint8, bboxes (n,4) as (y1,x1,y2,x2) -> the data-dict entries the
    model consumes (load_data_eval.py:253-288, 318-333, 361-377)."""
    im_H, im_W = image.shape[:2]
    coord_2d = get_2d_coord_ref(im_W, im_H)
    out = {k: [] for k in ("roi_img", "roi_mask", "roi_coord_2d", "roi_wh", "bbox_center", "resize_ratio")}
    for j, bbox in enumerate(bboxes):
        y1, x1, y2, x2 = [float(v) for v in bbox]                       # get_bbox_ori -> (rmin, rmax, cmin, cmax)
        bw = min(im_W, x2) - max(0, x1)                                 # get_real_hw (img_width=480 is the HEIGHT there)
        bh = min(im_H, y2) - max(0, y1)
        center = np.array([0.5 * (x1 + x2), 0.5 * (y1 + y2)])
        img_scale = max(y2 - y1, x2 - x1) * pad_scale
        img_scale = min(img_scale, max(im_H, im_W)) * 1.0
        M_img = get_affine_transform_ref(center, img_scale, img_size)
        M_out = get_affine_transform_ref(center, img_scale, out_res)
        roi = warp_affine_nearest_ref(image, M_img, img_size)
        roi = (roi / 255.0 - np.asarray(mean)) / np.asarray(std)
        out["roi_img"].append(roi.transpose(2, 0, 1))
        out["roi_coord_2d"].append(warp_affine_nearest_ref(coord_2d, M_out, out_res).transpose(2, 0, 1))
        m = masks[:, :, j].astype(np.float32)
        out["roi_mask"].append(warp_affine_nearest_ref(m, M_img, img_size)[None])
        out["roi_wh"].append(np.array([bw, bh], dtype=np.float32))
        out["resize_ratio"].append(out_res / img_scale)
        out["bbox_center"].append(center)
    return {k: np.asarray(v).astype(np.float32) for k, v in out.items()}


def pred_rt_ref(rot, trans, size, pred_scale):
    """evaluation/evaluate.py:116-125: pred_RT (B,4,4) and the L2-normalised size (torch F.normalize, eps 1e-12)."""
    rot, trans, size = np.asarray(rot, np.float32), np.asarray(trans, np.float32), np.asarray(size, np.float32)
    bs = rot.shape[0]
    n = np.maximum(np.sqrt((size * size).sum(1, keepdims=True, dtype=np.float32)), np.float32(1e-12))
    RT = np.zeros((bs, 4, 4), np.float32)
    RT[:, :3, :3] = rot
    RT[:, :3, 3] = trans
    RT[:, 3, 3] = 1
    RT[:, :3, :] = RT[:, :3, :] * np.asarray(pred_scale, np.float32)[:, None, None]
    return RT, (size / n).astype(np.float32)
