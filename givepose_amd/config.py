"""Explicit configuration for the PoseNet inference path.

The reference reads a global absl ``FLAGS`` singleton inside module constructors
(config/config.py:5-128; read sites: network/PoseNet.py:138-171,
network/conv_pnp_net.py:121,254, network/pose_head.py:22).  This dataclass carries
the ~12 flags the inference path reads, with the reference's defaults.
"""
from dataclasses import dataclass, field
from typing import Tuple


@dataclass(frozen=True)
class PoseNetConfig:
    main_backbone: str = "convnext"      # config.py:113 ('convnext' | 'resnet34' throughput variant)
    img_size: int = 256                  # config.py:20
    out_res: int = 64                    # config.py:21
    mask_attention_type: str = "none"    # config.py:22
    feat_ts: int = 128                   # config.py:39
    flat_op: str = "flatten"             # config.py:105
    t_type: str = "site"                 # config.py:108
    size_head_out_dim: int = 3           # config.py:109
    nocsmap_encoder: str = "conv"        # config.py:111 ('conv' | 'att')
    r_type: str = "allo_rot6d"           # config.py:116
    use_dcn: str = "dcnv3"               # config.py:120 ('dcnv3' | '')
    dataset: str = "CAMERA+Real"         # config.py:9 ('wild6d' rescales z, pose_from_pred_centroid_z.py:110)
    # ConvNeXt-Base (timm convnext_base, network/backbone.py:36-46)
    convnext_dims: Tuple[int, ...] = (128, 256, 512, 1024)
    convnext_depths: Tuple[int, ...] = (3, 3, 27, 3)
    # build-side switch (not a reference flag): fp16 stages with C in {128, 256} run fc1 -> GELU -> fc2 as one kernel
    fuse_mlp: bool = True
    # ... from this many crops per forward up: below it the fused kernel's few workgroups each walk both weight matrices and
    # the two plain GEMMs are the shorter chain.  Round 6 (scripts/fuse_mlp_sweep.py, profiles/r06_fuse_mlp_sweep.txt, with the round-5 four-wave
    # form of the C = 128 kernel): forward at B = 4 / 8 / 12: 2.30 / 2.81 / 3.17 ms fused against 2.24 / 2.76 / 3.16 as two GEMMs; B = 16 / 24 / 32 / 48:
    # 3.20 / 3.89 / 4.13 / 5.17 against 3.31 / 4.00 / 4.33 / 5.48 -> 16 (rounds 2-5: 32, fitted on the eight-wave kernel)
    fuse_mlp_min_batch: int = 16
    # build-side switch: fp16 stage with C = 512 runs the depth-wise conv one workgroup per 128-channel slab and applies
    # the block's LayerNorm in fc1's GEMM epilogue (algebraically identical: LN is affine per row).  Measured on MI355X
    # (bs 64): depth-wise 25.2 -> 22.9 us but fc1 52 -> 58 us per block, a net loss, hence off by default.
    defer_ln: bool = False

    # build-side switch (fp16 storage only): the residual stream of stage 2 (27 of the 36 ConvNeXt blocks) is accumulated in
    # fp32 -- the downsample conv and every fc2 epilogue write it in fp32 plus an fp16 copy for the depth-wise conv -- so that
    # the fp16 rounding of the stream is not compounded block after block.  Measured: DESIGN.md 5c.
    res_fp32: bool = False
    # build-side switch (fp16 storage only, round 5): the heads' ConvTranspose2d-as-GEMM writes its (B*64, 9*256) column matrix in fp16 (the GEMM's lean
    # epilogue) instead of fp32, and gp_deconv_col2im sums the fp16 summands in fp32: half the bytes of the two launches (75 -> 38 MB per head at
    # 128 crops); every summand is rounded to fp16 once, like every other activation of the mode.  Measured: +0.3 % end to end (docs/history/round5.md 8.4).
    deconv_cols_f16: bool = True
    # build-side switch (fp16 storage only, round 5): GroupNorm apply + GELU + the 1x1 out layer of the xyz heads on packed fp16 arithmetic (GP_ACT_PACKED16): the
    # per-channel scale and shift are rounded to fp16, so the absolute error grows like 2^-11 |group mean| / std (tests/test_hip_ops.py bounds it at ratios 5 .. 50)
    gnxyz16: bool = True
    # build-side switch (fp16 storage only, round 5): stage 2 (C = 512) runs fc1 -> GELU -> fc2 as ONE launch (convnext_mlp512_kernel) from 128 crops per launch up:
    # the 134 MB hidden tensor never exists (7.2 GB of HBM traffic per 128 crops), but the kernel alone is 8 % slower than the two launches (one wave per SIMD);
    # end to end +0.7 % in flight, -0.8 % serial (docs/history/round5.md 8.5).  Off by default.
    fuse_mlp512: bool = False

    @property
    def feature_channel(self) -> int:
        return {"convnext": self.convnext_dims[-1], "resnet34": 512}[self.main_backbone]
