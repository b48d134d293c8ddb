"""The per-frame body of the reference's evaluation loop (evaluation/evaluate.py:100-126) with every stage on the device:

    detections of a frame --gp_crop_rois--> roi_img / roi_mask / roi_coord_2d / roi_wh / bbox_center / resize_ratio
                          --Scale_net----> pred_scale                     (evaluate.py:112)
                          --PoseNet------> rot / trans / size             (evaluate.py:114)
                          --gp_pred_rt---> pred_RT (B,4,4), pred_size     (evaluate.py:116-125)

`FramePipeline` only strings the drop-in pieces together (SURVEY.md 8f rows 1-2 around row a1); it holds no arithmetic.
The full frame for Scale_net's second encoder (`full_img`, load_data_eval.py:336-338: the frame resized to 256x256 with
cv2.INTER_LINEAR when FLAGS.resize_full, normalised) is an INPUT here: cv2's bilinear resize is not restated.
"""
import numpy as np
import torch

from .postprocess import pred_rt
from .preprocess import RoiCropper


class FramePipeline:
    def __init__(self, network, scale_net, im_H=480, im_W=640, device="cuda", cats_num=6):
        self.net, self.scale_net, self.dev, self.cats = network, scale_net, torch.device(device), cats_num
        self.cropper = RoiCropper(im_H, im_W, self.dev, network.cfg.img_size, network.cfg.out_res)

    @torch.no_grad()
    def __call__(self, frame_u8, masks_u8, bboxes, cat_ids, cam_K, mean_shapes, full_img):
        """frame (H,W,3) uint8, masks (n,H,W) uint8, bboxes (n,4) (y1,x1,y2,x2), cat_ids (n,) 0-based, cam_K (3,3),
        mean_shapes (n,3) metres, full_img (3,S,S) fp32 normalised -> (pred_RT (n,4,4), pred_size (n,3), out dict), on the device."""
        n = len(bboxes)
        static = self.net.static_inputs(n, self.dev)
        self.cropper(torch.as_tensor(frame_u8)[None], masks_u8, [0] * n, list(range(n)), bboxes, out=static)
        static["cam_K"].copy_(torch.as_tensor(cam_K, dtype=torch.float32).expand(n, 3, 3))
        static["mean_size"].copy_(torch.as_tensor(mean_shapes, dtype=torch.float32))
        data = dict(static)
        data["full_img"] = torch.as_tensor(full_img, dtype=torch.float32).to(self.dev).expand(n, *full_img.shape[-3:]).contiguous()
        data["one_hot"] = torch.from_numpy(np.eye(self.cats, dtype=np.float32)[np.asarray(cat_ids)]).to(self.dev)   # evaluate.py:101-102
        pred_scale = self.scale_net(data, self.dev, "test")
        out = self.net.forward_device(static, self.dev)
        rt, size = pred_rt(out, pred_scale)
        return rt, size, out

    @torch.no_grad()
    def run_frames(self, frames_u8, masks_u8, bboxes, cat_ids, cam_K, mean_shapes, full_imgs):
        """The detections of SEVERAL frames in ONE launch sequence (the loop body of evaluation/evaluate.py:89-126 for F frames at once;
        every frame keeps the DCNv3 prefix coupling of its own `forward`: PoseNet.forward_device(groups=...)).
        frames (F,H,W,3) uint8; masks_u8 / bboxes / cat_ids / mean_shapes: per-frame lists (n_f,H,W) / (n_f,4) / (n_f,) / (n_f,3); cam_K (3,3)
        or (F,3,3); full_imgs (F,3,S,S).  Returns (pred_RT (N,4,4), pred_size (N,3), out dict, sizes) with N = sum n_f, frame-major."""
        sizes = [len(b) for b in bboxes]
        n, F = sum(sizes), len(sizes)
        static = self.net.static_inputs(n, self.dev, ragged=True)
        frame_of = [f for f, k in enumerate(sizes) for _ in range(k)]
        allmasks = torch.cat([torch.as_tensor(m) for m in masks_u8], 0)
        allboxes = np.concatenate([np.asarray(b).reshape(-1, 4) for b in bboxes], 0)
        self.cropper(torch.as_tensor(frames_u8), allmasks, frame_of, list(range(n)), allboxes, out=static)
        K = torch.as_tensor(cam_K, dtype=torch.float32)
        static["cam_K"].copy_(K.expand(n, 3, 3) if K.dim() == 2 else K[frame_of])
        static["mean_size"].copy_(torch.cat([torch.as_tensor(m, dtype=torch.float32).reshape(-1, 3) for m in mean_shapes], 0))
        data = dict(static)
        data["full_img"] = torch.as_tensor(full_imgs, dtype=torch.float32).to(self.dev)[frame_of].contiguous()
        cats = np.concatenate([np.asarray(c).reshape(-1) for c in cat_ids])
        data["one_hot"] = torch.from_numpy(np.eye(self.cats, dtype=np.float32)[cats]).to(self.dev)
        pred_scale = self.scale_net(data, self.dev, "test")
        out = self.net.forward_device(static, self.dev, groups=sizes)
        rt, size = pred_rt(out, pred_scale)
        return rt, size, out, sizes
