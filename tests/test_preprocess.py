"""Crop pre-processing: oracle properties on CPU, HIP kernel vs oracle on the GPU (bit-exact)."""
import numpy as np
import pytest
import torch

from oracle import preprocess_ref as O


def _scene(seed, n=5, H=480, W=640):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    masks = rng.random((H, W, n)) > 0.5
    b = []
    for _ in range(n):
        y1, x1 = rng.integers(-30, H - 60), rng.integers(-30, W - 60)
        b.append([y1, x1, y1 + rng.integers(24, 300), x1 + rng.integers(24, 300)])
    b[-1] = [0, 0, H, W]                        # box larger than the clip: scale = max(H, W)
    return img, masks, np.array(b)


def test_oracle_identity_and_border():
    """A crop whose source square equals its output size is a pure integer shift; outside the frame is 0."""
    img = np.arange(40 * 50 * 3, dtype=np.uint8).reshape(40, 50, 3)
    M = O.get_affine_transform_ref([20.0, 16.0], 16.0, 16)       # 16-px square centred at (20,16) -> 16x16
    out = O.warp_affine_nearest_ref(img, M, 16)
    assert np.array_equal(out, img[8:24, 12:28])
    M = O.get_affine_transform_ref([2.0, 2.0], 16.0, 16)         # hangs over the top-left corner
    out = O.warp_affine_nearest_ref(img, M, 16)
    assert np.array_equal(out[6:, 6:], img[:10, :10]) and not out[:6].any() and not out[:, :6].any()


def test_oracle_matches_float_nearest_away_from_ties():
    """The fixed-point walk equals round-to-nearest of the exact inverse map except within 2^-10 px of a tie."""
    img, _, boxes = _scene(3)
    for y1, x1, y2, x2 in boxes[:3]:
        c = np.array([0.5 * (x1 + x2), 0.5 * (y1 + y2)])
        sc = min(max(y2 - y1, x2 - x1) * 1.5, 640.0)
        M = O.get_affine_transform_ref(c, sc, 64)
        iM = O.invert_affine_ref(M)
        xs, ys = np.meshgrid(np.arange(64.0), np.arange(64.0))
        fx, fy = iM[0] * xs + iM[1] * ys + iM[2], iM[3] * xs + iM[4] * ys + iM[5]
        safe = (np.abs(fx - np.floor(fx) - 0.5) > 4e-3) & (np.abs(fy - np.floor(fy) - 0.5) > 4e-3)
        X, Y = np.floor(fx + 0.5).astype(int), np.floor(fy + 0.5).astype(int)
        ok = (X >= 0) & (X < 640) & (Y >= 0) & (Y < 480)
        ref = np.where(ok[..., None], img[np.clip(Y, 0, 479), np.clip(X, 0, 639)], 0)
        got = O.warp_affine_nearest_ref(img, M, 64)
        assert np.array_equal(got[safe], ref[safe])


def test_host_params_match_oracle():
    from givepose_amd import preprocess as P
    img, masks, boxes = _scene(4)
    d = O.crop_batch_ref(img, masks, boxes)
    p = P.crop_params(boxes, 480, 640)
    for k in ("roi_wh", "bbox_center", "resize_ratio"):
        assert np.array_equal(p[k], d[k]), k
    il, xl, yl = P.luts(480, 640)
    g = O.get_2d_coord_ref(640, 480)
    assert np.array_equal(xl, g[0, :, 0]) and np.array_equal(yl, g[:, 0, 1])
    v = np.arange(256)[:, None] * np.ones((1, 3))
    assert np.array_equal(il.T, ((v / 255.0 - np.asarray(P.IMG_MEAN)) / np.asarray(P.IMG_STD)).astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(5, 1), (6, 7), (7, 64)])
def test_crop_rois_hip_bit_exact(seed, n):
    from givepose_amd import preprocess as P
    dev = torch.device("cuda:0")
    frames, masks, fidx, midx, boxes, ref = [], [], [], [], [], {}
    per = [min(4, n - i) for i in range(0, n, 4)]             # up to 4 detections per frame, several frames
    for f, k in enumerate(per):
        img, m, b = _scene(seed * 100 + f, n=k)
        d = O.crop_batch_ref(img, m, b)
        for key, v in d.items():
            ref.setdefault(key, []).append(v)
        frames.append(img)
        for j in range(k):
            fidx.append(f)
            midx.append(len(masks))
            masks.append(m[:, :, j].astype(np.uint8))
            boxes.append(b[j])
    ref = {k: np.concatenate(v) for k, v in ref.items()}
    crop = P.RoiCropper(480, 640, dev)
    out = crop(np.stack(frames), np.stack(masks), fidx, midx, np.array(boxes))
    torch.cuda.synchronize()
    for k, v in ref.items():
        got = out[k].cpu().numpy()
        assert got.shape == v.shape, k
        assert np.array_equal(got, v), (k, float(np.abs(got - v).max()))


@pytest.mark.gpu
def test_crop_rois_feeds_posenet_inputs_and_rejects_bad_indices():
    from givepose_amd import preprocess as P
    dev = torch.device("cuda:0")
    img, m, b = _scene(9, n=3)
    crop = P.RoiCropper(480, 640, dev)
    with pytest.raises(ValueError):
        crop(img[None], np.moveaxis(m, 2, 0).astype(np.uint8), [0, 0, 1], [0, 1, 2], b)      # frame 1 does not exist
    with pytest.raises(ValueError):
        crop(img[None], np.moveaxis(m, 2, 0).astype(np.uint8), [0, 0, 0], [0, 1, 3], b)      # mask 3 does not exist
    static = {"roi_img": torch.zeros(3, 3, 256, 256, device=dev), "roi_mask": torch.zeros(3, 1, 256, 256, device=dev)}
    out = crop(img[None], np.moveaxis(m, 2, 0).astype(np.uint8), [0, 0, 0], [0, 1, 2], b, out=static)
    assert out["roi_img"] is static["roi_img"] and float(static["roi_img"].abs().sum()) > 0
    with pytest.raises(RuntimeError):
        P.RoiCropper(480, 640, "cpu")


@pytest.mark.gpu
def test_pred_rt_hip_vs_oracle():
    from givepose_amd import postprocess as PP
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    B = 37
    out = {"rot": torch.randn(B, 3, 3, generator=g), "trans": torch.randn(B, 3, generator=g), "size": torch.randn(B, 3, generator=g) * 0.2}
    out["size"][3] = 0.0                                    # F.normalize eps branch
    sc = torch.rand(B, generator=g) + 0.5
    ref_rt, ref_ps = O.pred_rt_ref(out["rot"].numpy(), out["trans"].numpy(), out["size"].numpy(), sc.numpy())
    rt, ps = PP.pred_rt({k: v.to(dev) for k, v in out.items()}, sc)
    assert np.array_equal(rt.cpu().numpy(), ref_rt)                                  # products of two fp32: exact
    assert np.abs(ps.cpu().numpy() - ref_ps).max() <= 2e-7                           # fp32 sqrt / divide, 1 ulp
    rt1, _ = PP.pred_rt({k: v.to(dev) for k, v in out.items()})
    assert np.array_equal(rt1.cpu().numpy()[:, :3, :3], out["rot"].numpy())
    with pytest.raises(RuntimeError):
        PP.pred_rt(out)


def test_vectorised_crop_params_equal_the_per_detection_form():
    """crop_params (batched numpy, what the product runs per frame) against the per-detection form it was vectorised from:
    bit for bit -- the inverse maps feed the kernel's fixed-point rounding."""
    from givepose_amd import preprocess as P
    rng = np.random.default_rng(3)
    n = 300
    y1, x1 = rng.integers(0, 200, n), rng.integers(0, 300, n)
    b = np.stack([y1, x1, y1 + rng.integers(10, 400, n), x1 + rng.integers(10, 500, n)], 1).astype(np.float64)
    b[:100] += rng.random((100, 4))
    A, B = P.crop_params(b, 480, 640), P.crop_params_loop(b, 480, 640)
    for k in A:
        assert A[k].dtype == B[k].dtype and A[k].shape == B[k].shape and np.array_equal(A[k], B[k]), k
