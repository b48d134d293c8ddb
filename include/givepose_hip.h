/* givepose_hip.h -- C ABI of libgivepose_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the GIVEPose PoseNet inference path.  The one
 * native op of the reference on this path is the pybind module `DCNv3`
 * (network/ops_dcnv3/src/vision.cpp:14-17, dcnv3.h:20-38): gp_dcnv3_forward replaces
 * `dcnv3_forward`.  Every other entry point replaces an ATen/cuDNN call the reference makes from
 * Python on this path; the replaced call site is cited on each.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless named host_*; the caller owns every buffer
 *     (outputs are caller-allocated, no hidden allocation, no hidden synchronisation);
 *   - activations are channels-last: (N, H, W, C) contiguous, C fastest;
 *   - `dtype` is the storage type of activations and weights: GP_F32 or GP_F16; accumulation,
 *     normalisation statistics, biases and norm affine parameters are always fp32;
 *   - `stream` is a hipStream_t (0 = the null stream); launches are asynchronous;
 *   - return value 0 on success, negative gp_status otherwise (never printf-and-continue, unlike
 *     dcnv3_im2col_cuda.cuh:913-916); gp_last_error() gives the message for the calling thread.
 */
#ifndef GIVEPOSE_HIP_H
#define GIVEPOSE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

enum gp_status { GP_OK = 0, GP_ERR_INVALID = -1, GP_ERR_LAUNCH = -2, GP_ERR_RUNTIME = -3 };
enum gp_dtype { GP_F32 = 0, GP_F16 = 1, GP_F64 = 2 /* gp_dcnv3_forward_any / gp_dcnv3_backward only */ };
/* OR-ed into `dtype` = GP_F32 of gp_dwconv_ln / gp_layernorm / gp_groupnorm_apply / gp_upsample_bilinear2x: the (dense) output is
 * written as the two fp16 planes of the split-operand GEMM mode (hi at y, lo' = fp16((v - hi) * 2^GP_SPLIT_SHIFT) one tensor's
 * worth of elements behind it, in y's fp32-sized storage) instead of fp32 -- the producer emits what gp_split_planes would make
 * of its result, and the consumer is gp_gemm(split_shift = GP_SPLIT_SHIFT) reading X = y.  y must not alias x. */
#define GP_OUT_PLANES 0x100
/* OR-ed into `dtype` = GP_F16 of gp_layernorm: the input rows are fp32 (the fp32 residual stream of the fp16 mode), the output fp16 */
#define GP_IN_F32 0x200
#define GP_SPLIT_SHIFT 11
enum gp_act { GP_ACT_NONE = 0, GP_ACT_GELU = 1, GP_ACT_RELU = 2, GP_ACT_LRELU = 3 /* slope 0.1 */ };
/* GEMM epilogues: v = acc + bias; then */
enum gp_epilogue {
    GP_EPI_NONE = 0,      /* out = v                                        */
    GP_EPI_GELU = 1,      /* out = gelu_erf(v)                              */
    GP_EPI_RELU = 2,      /* out = max(v, 0)                                */
    GP_EPI_LRELU = 3,     /* out = v > 0 ? v : 0.1 v                        */
    GP_EPI_SCALE_RES = 4, /* out = residual + gamma[n] * v  (ConvNeXt block) */
    GP_EPI_RES_RELU = 5,  /* out = max(residual + v, 0)    (ResNet BasicBlock, network/resnet.py:49-52) */
    /* LayerNorm of the X rows folded into the epilogue (fp16, large-tile kernels): with X = un-normalised rows,
     * W = fc.weight * ln.weight[k], ln_colsum[n] = sum_k W[n][k], bias[n] = fc.weight @ ln.bias + fc.bias and per-row
     * (mean, rstd) from ln_stats (gp_dwconv7_raw_stats):  out = gelu( rstd[m] * (acc - mean[m] * ln_colsum[n]) + bias[n] ),
     * which equals gelu(fc(LayerNorm(x))) -- ConvNeXt block norm -> mlp.fc1 -> act */
    GP_EPI_LNFOLD_GELU = 6
};

const char* gp_last_error(void);
#define GP_ABI_VERSION 323 /* round 6: + gp_dwconv_ln_groups (322), gp_convnext_mlp_pack_w2_s32 / GP_MLP_S32 (323); 321 = round 5: gp_gemm_desc.gn_rows + gp_gemm_gn_rows (321); gp_gemm variants 19-22, gp_convnext_mlp C = 512 (no layout change); 311 = round 4 (+ gp_groupnorm_upsample2x); 310 = round 3 (gp_gemm_desc: split-operand / fp32 residual stream fields); 200 = round 2 */
int gp_version(void);   /* == GP_ABI_VERSION of the header the library was built from */
/* device properties the host needs: CU count and arch string ("gfx950...") */
int gp_device_info(int* cu_count, char* arch, int arch_len);

/* ---------------------------------------------------------------------------------------------
 * DCNv3 forward -- replaces DCNv3.dcnv3_forward (network/ops_dcnv3/src/dcnv3.h:20-38 ->
 * cuda/dcnv3_cuda.cu:21-85 -> cuda/dcnv3_im2col_cuda.cuh:216-282).
 *   in   (N,H,W,G*D) dtype;  out (N,Ho,Wo,G*D) dtype, Ho = (H+2*pad-(dil*(K-1)+1))/stride+1.
 *   offset / mask: FLAT buffers of om_dtype indexed exactly as the CUDA kernel does, i.e. row
 *     r = (b*Ho+ho)*Wo+wo, element r*off_ld + (g*P+p)*2 + {0:w,1:h} and r*mask_ld + g*P+p, with
 *     P = K*K - remove_center and taps ordered kernel_w outer / kernel_h inner.  With
 *     off_ld = G*P*2 and mask_ld = G*P this is the reference addressing, including the stride-2
 *     case where the buffers are (N,H,W,..)-shaped and only the flat prefix is consumed.
 *   mask_is_logits != 0: `mask` holds pre-softmax logits and the kernel applies the softmax over
 *     the P taps of each group (fuses modules/dcnv3.py:331-333 into the gather).
 *   Requires batch <= im2col_step or batch % im2col_step == 0 like dcnv3_cuda.cu:46-49.
 */
int gp_dcnv3_forward(const void* in, const void* offset, const void* mask, void* out, int N, int H, int W,
                     int G, int D, int K, int stride, int pad, int dil, float offset_scale,
                     int remove_center, int im2col_step, int off_ld, int mask_ld, int mask_is_logits,
                     int dtype, int om_dtype, void* stream);

/* The same operator at the reference's full breadth: the argument list of DCNv3.dcnv3_forward itself
 * (dcnv3.h:20-38: kernel / stride / pad / dilation per axis, any group_channels) and its dtype dispatch
 * (dcnv3_cuda.cu:68: GP_F64 / GP_F32 / GP_F16; opmath = double for double, float otherwise).  offset / mask are
 * contiguous flat buffers of `dtype` with the reference addressing (off_ld = G*P*2, mask_ld = G*P). */
int gp_dcnv3_forward_any(const void* in, const void* offset, const void* mask, void* out, int N, int H, int W, int G,
                         int D, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                         int dilation_h, int dilation_w, float offset_scale, int remove_center, int im2col_step,
                         int dtype, void* stream);

/* DCNv3 backward -- replaces DCNv3.dcnv3_backward (network/ops_dcnv3/src/dcnv3.h:40-59 -> cuda/dcnv3_cuda.cu:87-174 ->
 * cuda/dcnv3_im2col_cuda.cuh:386-487 + :82-140).  grad_out (N,Ho,Wo,G*D) `dtype`; grad_in (N,H,W,G*D), grad_offset
 * (offset_numel), grad_mask (mask_numel) are OPMATH typed like the reference's (float for GP_F16 / GP_F32, double for
 * GP_F64; dcnv3_cuda.cu:123-130) and are zero-filled here before the accumulation, so the tail of an offset buffer
 * longer than the consumed prefix stays zero.  grad_in is accumulated with atomics (run-to-run rounding differences in
 * the last bits, as in the reference); grad_offset / grad_mask have one writer per element. */
int gp_dcnv3_backward(const void* in, const void* offset, const void* mask, const void* grad_out, void* grad_in,
                      void* grad_offset, void* grad_mask, long offset_numel, long mask_numel, int N, int H, int W, int G,
                      int D, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                      int dilation_h, int dilation_w, float offset_scale, int remove_center, int im2col_step,
                      int dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused GEMM / implicit-GEMM convolution on MFMA.
 *   C[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] ),  m < M, n < N.
 * Replaces F.linear / nn.Conv2d (1x1, 2x2 s2, 3x3 s1/s2) call sites: timm ConvNeXt pointwise
 * MLPs + downsample convs (network/backbone.py:36-46), ConvModule convs
 * (network/torch_utils/layers/conv_module.py:224-234), ConvPnPNet convs/fc
 * (network/conv_pnp_net.py:164-199), DCNv3 Linear projections (ops_dcnv3/modules/dcnv3.py:325-354),
 * feat_reducer (network/PoseNet.py:192), ConvTranspose2d as GEMM + gp_deconv_col2im.
 *   W: (N, K) row-major, K fastest; for conv mode K index = (kh*KW + kw)*Cin + ci.
 *   conv mode (KH > 0): X is (B,H,W,Cin) channels-last, M = B*Ho*Wo, zero padding `pad`;
 *     requires Cin % (128/sizeof(dtype)) == 0.  plain mode (KH == 0): X is (M, K) with row stride ldx.
 *   K % (128/sizeof(dtype)) == 0 and N % 4 == 0 required.
 *   bias/gamma fp32 [N] (bias may be NULL); residual same dtype as X, row stride ldres.
 *   out_f32 != 0 stores fp32 instead of dtype.  C row stride ldc (elements of the output type).
 *   splitk > 1: K is cut in `splitk` slices; `workspace` must hold splitk*M*N floats.
 */
typedef struct gp_gemm_desc {
    const void* X;
    const void* W;
    const float* bias;
    const float* gamma;
    const void* residual;
    void* C;
    float* workspace;
    int M, N, K;
    int ldx, ldc, ldres;
    int epilogue;
    int out_f32;
    int splitk;
    /* conv mode */
    int B, H, Win, Cin, KH, KW, stride, pad, Ho, Wo;
    int dtype;
    /* optional fused GroupNorm statistics of the OUTPUT (large-tile variants and variant 18, no split-K): per 64 output rows and
     * channel group (sum, sum of squares) -> gn_partial (M/64, gn_groups, 2) fp32, i.e. (B, HW/64, G, 2) when
     * M = B*gn_hw; consumed by gp_groupnorm_apply(..., chunks = gn_hw/64). NULL = off.  (gn_rows below: other chunk sizes.) */
    float* gn_partial;
    int gn_groups, gn_hw;
    int variant; /* 0 = choose by shape (the product path); otherwise one schedule, for tests and A/B runs: 4 = 128x128 LDS-DMA
                  * tile (the split-K carrier), 5 / 9 = its 4-stage forms, 7 = 128x128 software-pipelined, two workgroups per
                  * CU, 2 / 8 = 256x128, 3 = 256x256, 10 / 11 / 12 = ping-pong 256x256 / 128x256 / 5-stage, 13 = 3x3 window
                  * conv (Cout 256), 16 / 17 = K 512 with the weight slice resident in registers (17: 16-byte stores, needs ldc % 8 == 0 and a
                  * 16-byte aligned C; the default of stage-2 fc1 until round 4), 19 / 20 / 21 / 22 = 17's arithmetic with two accumulator sets (round 5): 19 / 20 on
                  * 32x32x16 MFMAs, 22 / 21 on 16x16x32; 20 / 21 apply GP_EPI_GELU on packed fp16 (13 v_pk_* operations per value pair, absolute error of
                  * one fp16 rounding: docs/history/round5.md 8.2; refused for the other epilogues); 21 = the default of stage-2 fc1 (22 with GP_GELU16=0),
                  * 18 = the small-M latency kernel (few rows -- the detections of one
                  * frame: fp16 in / out, N % 32 == 0, M % 16 == 0 (% 64 with gn_partial), plain GEMM or conv; chosen by variant 0 when its
                  * estimate beats the tile kernels'; a split-K request is ignored; 218 / 318 / 418 force the 16 / 32 / 64-row tile),
                  * 23 = the row-vector kernel (round 5: a plain fp16 GEMM of M <= 8 rows, K % 512 == 0, N % 8 == 0, bias / GELU / ReLU / LeakyReLU -- ConvPnPNet's fc
                  * layers over the detections of one frame; chosen by variant 0 for such shapes, GP_GEMM_GEMV=0 keeps it out; a split-K request is ignored).
                  * + 100 n otherwise: timing ablations. */
    /* GP_EPI_LNFOLD_GELU only: ln_stats (M, 2, ln_nslab) fp32 partial (sum, sum of squares) of each X row over
     * ln_nslab channel slabs; ln_colsum (N) fp32; ln_eps.  Requires M % 256 == 0, N % 256 == 0, fp16 output. */
    const float* ln_stats;
    const float* ln_colsum;
    int ln_nslab;
    float ln_eps;
    /* != 0: the caller keeps other launches running beside this one (independent batches in flight), so the tile is
     * chosen for cost per FLOP rather than for filling 256 CUs alone (variant == 0 only) */
    int co_scheduled;
    /* optional hint: `prefetch_bytes` bytes at `prefetch` (the weights of the launch that FOLLOWS on this stream) are
     * touched by this launch's workgroups as they start, so that they are in L2 / Infinity Cache when the next kernel
     * wants them (the ~230 MB of weights of a step do not survive a step in the caches; a 128x128-tile GEMM whose
     * weights arrive from HBM runs 10-25 % longer).  Speed only, never dereferenced for a result.  NULL / 0 = off.
     * Honoured by every variant. */
    const void* prefetch;
    long prefetch_bytes;
    /* split-operand mode (parity-grade results at the fp16 MFMA rate), split_shift = S > 0:
     *   X and W point at fp16 PLANES: hi = fp16(v) at the pointer, lo' = fp16((v - hi) * 2^S) x_plane_stride / w_plane_stride
     *   ELEMENTS behind it (gp_split_planes makes them; weights are split once by the host); the kernel accumulates
     *   x_hi w_lo' + x_lo' w_hi, scales by 2^-S (exact) and adds x_hi w_hi, all in fp32: |error| ~ 2^-22 |x||w| per product,
     *   i.e. what an fp32 GEMM's own accumulation rounding amounts to.  Requires dtype GP_F16, out_f32 != 0 (C fp32);
     *   residual (if any) is fp32; ldx / Cin / K address one plane; variant 0 / 4 / 7 / 8 / 10 / 13 (the 3x3 window conv). */
    int split_shift;
    long x_plane_stride, w_plane_stride;
    /* split-operand mode only, out_planes != 0: C is written as fp16 planes (hi at C, lo' c_plane_stride elements behind it,
     * row stride ldc in fp16 elements) -- the X operand of the next split-operand gp_gemm -- instead of fp32; no split-K. */
    int out_planes;
    long c_plane_stride;
    /* fp32 residual stream of the fp16 mode (dtype GP_F16, out_f32 != 0): residual_f32 != 0: `residual` is fp32 (row stride ldres
     * in fp32 elements) -- variants 0 / 7 / 10; c16 != NULL: the output values are ALSO stored rounded to fp16 at c16 (row stride
     * ldc16): the stream is accumulated in fp32 in C while the next consumer (depth-wise conv, LayerNorm) reads the fp16 copy. */
    int residual_f32;
    void* c16;
    int ldc16;
    /* fused GroupNorm statistics (gn_partial != NULL): rows per statistics chunk, 0 = 64.  16 / 32: gn_partial is (M/gn_rows, gn_groups, 2) and the
     * consumers take chunks = gn_hw/gn_rows; the small-M kernel (variant 18) only -- gp_gemm fails where it cannot take the launch.  Ask
     * gp_gemm_gn_rows() which value to use for a shape. */
    int gn_rows;
} gp_gemm_desc;
/* Rows per statistics chunk the library wants for a fused-GroupNorm fp16 GEMM / conv of M rows (M = B * hw), N columns, K: 64 (the tile kernels'
 * chunk) or, where the small-M latency kernel takes the launch and a smaller tile is faster by its cost model, 32 / 16 (round 5: the heads' 3x3 convs
 * of network/xyz_head.py:241-316 at 1-4 crops).  Pure function of the shape and of GP_GEMM_SMALLM. */
int gp_gemm_gn_rows(int M, int N, int K, int hw);
int gp_gemm(const gp_gemm_desc* d, void* stream);

/* fp32 rows -> the two fp16 planes of the split-operand mode: hi (rows, cols) at `planes`, lo' plane_stride elements behind
 * it; x rows have stride ldx (elements), the planes are dense (row stride = cols).  cols % 8 == 0, x 16-byte aligned rows. */
int gp_split_planes(const float* x, void* planes, long rows, int cols, long ldx, long plane_stride, int split_shift,
                    void* stream);

/* Fused ConvNeXt block MLP (fp16 storage, C = 128, 256 or -- round 5, one wave per SIMD, M % 128 == 0, needs the packed-fp16 GELU
 * (GP_GELU16 != 0), measured slower than gp_gemm x 2 at 128 crops: profiles/r05_mlp512_ab.txt -- 512): one launch for
 *   out = residual + gamma * ( fc2( GELU( fc1(x) ) ) )
 * i.e. timm ConvNeXtBlock.forward's `mlp` + layer scale + shortcut (built by network/backbone.py:36-46); replaces the
 * gp_gemm(GELU) -> gp_gemm(SCALE_RES) pair for the stages whose 4C-wide hidden tensor would otherwise round-trip HBM.
 *   x (M,C) = LayerNorm output, w1 (4C,C), b1 (4C) fp32, b2/gamma (C) fp32, residual/out (M,C) (out may alias
 *   residual, not x); w2p = fc2.weight (C,4C) re-ordered by gp_convnext_mlp_pack_w2 (k-slot order of the MFMA B
 *   fragment that the GELU output forms in registers).  M % 256 == 0; all pointers 16-byte aligned.
 *   The GELU runs on packed fp16 arithmetic (common.hpp gelu16_slice) unless the environment has GP_GELU16=0. */
int gp_convnext_mlp_pack_w2(const void* w2, void* w2p, int C, void* stream);
/* round 6: the column order of the 32x32x16-MFMA form of gp_convnext_mlp (C = 128 / 256): pass GP_F16 | GP_MLP_S32 as `dtype` together with it */
int gp_convnext_mlp_pack_w2_s32(const void* w2, void* w2p, int C, void* stream);
#define GP_MLP_S32 0x400
int gp_convnext_mlp(const void* x, const void* w1, const float* b1, const void* w2p, const float* b2,
                    const float* gamma, const void* residual, void* out, long M, int C, int dtype, void* stream);

/* ConvNeXt stem: conv4x4 s4 (3->C0, bias) + LayerNorm over channels (eps).  img is the
 * reference's NCHW fp32 `roi_img` (network/PoseNet.py:174); out is (B,H/4,W/4,C0) channels-last.
 * w is (48, C0) fp32 tap-major, k = c*16 + kh*4 + kw (the checkpoint's (C0,3,4,4) transposed by the host).
 * C0 must be 128. */
int gp_convnext_stem(const float* img, const float* w, const float* b, const float* ln_w, const float* ln_b,
                     void* out, int B, int H, int W, int C0, float eps, int dtype, void* stream);

/* depth-wise KSxKS conv (pad KS/2, stride 1, bias) + LayerNorm over C (+ optional GELU):
 * ConvNeXt block front half (dw7x7 -> LN) and DCNv3's dw_conv branch (dw3x3 -> LN -> GELU,
 * ops_dcnv3/modules/dcnv3.py:277-296).  wt: (KS*KS, C) tap-major, of `dtype`.  Only the first
 * `n_pixels` flat pixels (b,h,w order) are produced (DCNv3 consumes a prefix).
 * `act`: a gp_act code.  Codes >= 100 are TEST HOOKS that pin the kernel form the routing would otherwise pick by grid size (the parity tests compare the forms
 * with each other; results within the tolerances documented there, or the same bits where stated): 104 / 107 LDS-tiled VALU / strip kernel (KS = 7), 110 / 112
 * the 16 x 8 / 16 x 4 tiles of dwconv7_ln_tall_kernel, 113 / 114 its pair tiles of 4 / 2 rows (C = 1024, 8 x 8 maps), 120 + act / 125 + act the 16 x 4 / 16 x 2
 * tiles of dwconv3_ln_tile_kernel (KS = 3, C = 256, act = GELU: on another shape GP_ERR_INVALID; the other codes fall back to the routing).  Not part of the stable ABI. */
int gp_dwconv_ln(const void* x, const void* wt, const float* bias, const float* ln_w, const float* ln_b,
                 void* y, int B, int H, int W, int C, int KS, float eps, int act, long n_pixels, int dtype,
                 void* stream);

/* The same for SEVERAL batches in one launch (the detections of several frames, each frame one `PoseNet.forward` of the reference:
 * evaluation/evaluate.py:89-114): y (B*H*W/4, C) = for every crop the quarter-size flat prefix rows DCNv3 consumes
 * (ops_dcnv3/modules/dcnv3.py:318-356 hands the CUDA kernel offsets of an (N,H,W) grid that it reads as (N,H/2,W/2): SURVEY.md 0.3),
 * where output row j of a crop whose batch starts at crop crop_group_start[crop] is the result at flat full-resolution pixel
 * j + 3 * crop_group_start[crop] * H*W/4 of x -- i.e. every batch reads the prefix of ITS OWN flat pixel list.  crop_group_start:
 * B int32 on the device (one table entry per crop; entries are clamped to [0, crop]).  W % 16 == 0, H even, y != x. */
int gp_dwconv_ln_groups(const void* x, const void* wt, const float* bias, const float* ln_w, const float* ln_b, void* y, int B,
                        int H, int W, int C, int KS, float eps, int act, const int* crop_group_start, int dtype, void* stream);

/* ConvNeXt block front half with the LayerNorm deferred to the consuming GEMM (GP_EPI_LNFOLD_GELU): depth-wise 7x7
 * (pad 3, bias) only; y = conv output rounded to fp16, stats (B*H*W, 2, C/128) fp32 = per pixel the (sum, sum of
 * squares) of those rounded values over each 128-channel slab.  One workgroup per (16x4 pixel tile, slab), so the
 * launch has C/128 times the workgroups of gp_dwconv_ln.  fp16, C % 128 == 0, H % 4 == 0, W % 16 == 0. */
int gp_dwconv7_raw_stats(const void* x, const void* wt, const float* bias, void* y, float* stats, int B, int H, int W,
                         int C, int dtype, void* stream);

/* row LayerNorm over C (ConvNeXt downsample LayerNorm2d, ViT-block norms); y row stride ldy (0 = C). */
int gp_layernorm(const void* x, const float* w, const float* b, void* y, long rows, int C, float eps, int ldy,
                 int dtype, void* stream);

/* GroupNorm (nn.GroupNorm(G, C), eps) over channels-last x (B, HW, C):
 *   gp_groupnorm_stats -> partial (B, chunks, G, 2) fp32 = per-chunk (sum, sum of squares), chunks =
 *   gp_groupnorm_chunks(B, HW), summed in a fixed order (bitwise reproducible);
 *   gp_groupnorm_apply: finalises (mean, rstd) from `partial`, y = act((x-mean)*rstd*w + b); y row stride ldy
 *   (concat targets); in-place (y == x, ldy == C) allowed. */
int gp_groupnorm_chunks(int B, int HW);
int gp_groupnorm_stats(const void* x, float* partial, int B, int HW, int C, int G, int dtype, void* stream);
int gp_groupnorm_apply(const void* x, const float* partial, const float* w, const float* b, void* y, int B,
                       int HW, int C, int G, float eps, int act, int ldy, int chunks /* 0 = gp_groupnorm_chunks */,
                       int dtype, void* stream);

/* GroupNorm apply (statistics from `partial`: gp_groupnorm_stats, or the fused statistics of the producing gp_gemm) + activation +
 * bilinear x2 upsample (align_corners = True) in ONE pass: x (B, H, W, C) fp16 -> y (B, 2H, 2W, C) fp16.  Replaces ConvModule's
 * norm + act followed by nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True) of TopDownXyzHead
 * (network/xyz_head.py:250-264, :349-366); bitwise gp_groupnorm_apply followed by gp_upsample_bilinear2x.  fp16 only. */
int gp_groupnorm_upsample2x(const void* x, const float* partial, const float* w, const float* b, void* y, int B, int H, int W,
                            int C, int G, float eps, int act, int chunks, int dtype, void* stream);

/* gp_groupnorm_apply fused with gp_xyz_out_layer (the normalised tensor is consumed only by the 1x1 out layer and is
 * never written): out_w (3,C), out_b (3) fp32; outputs as gp_xyz_out_layer.  act | GP_ACT_PACKED16 (fp16, C = 256, GELU; round 5): the affine and the GELU
 * on packed fp16 arithmetic (13 operations per value pair instead of ~34; outputs within ~3e-4 mean / 2e-3 max of the fp32-accurate form, which is what the
 * fp16 mode's storage rounding amounts to anyway); ignored for the other dtypes / shapes. */
#define GP_ACT_PACKED16 0x100
int gp_groupnorm_apply_xyz(const void* x, const float* partial, const float* w, const float* b, const float* out_w,
                           const float* out_b, float* out_nchw, float* out_nhwc4, int B, int HW, int C, int G,
                           float eps, int act, int chunks, int dtype, void* stream);

/* nn.UpsamplingBilinear2d(scale_factor=2) (align_corners=True), channels-last (xyz_head.py:264). */
int gp_upsample_bilinear2x(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);

/* ConvTranspose2d(k3,s2,p1,op1,bias=False) second half: cols (B*H*W, 9*C) fp32 from gp_gemm with
 * W[(kh*3+kw)*C + co][ci] -> out (B,2H,2W,C) dtype (xyz_head.py:250-259).  dtype = GP_F16 | GP_COLS_F16 (round 5): cols are fp16
 * (the GEMM's lean fp16 epilogue: half the bytes; every summand rounded to fp16 once, summed in fp32; C % 8 == 0). */
#define GP_COLS_F16 0x400
int gp_deconv_col2im(const void* cols, void* out, int B, int H, int W, int C, int dtype, void* stream);

/* xyz out layer: Conv2d(C,3,1)+bias (xyz_head.py:317-324).  Writes the reference-layout NCHW fp32 map
 * (B,3,HW) and a channels-last (B*HW, 4) fp32 copy (x,y,z,0) for the next consumer. */
int gp_xyz_out_layer(const void* x, const float* w, const float* b, float* out_nchw, float* out_nhwc4,
                     int B, int HW, int C, int dtype, void* stream);

/* Conv2d(3, Cout, 1)+bias on the (B*HW,4) fp32 coordinate map (DCNv3_C.conv of the first MAPEncoder
 * layer, network/dcnv3.py:26,33). w (Cout,3) fp32. */
int gp_pointwise_k3(const float* xyz4, const float* w, const float* b, void* y, long rows, int Cout,
                    int dtype, void* stream);

/* ConvPnPNet first conv: Conv2d(5,Cout,3,s2,p1,bias=False) on cat(ivfc (B*HW,4) fp32, roi_coord_2d
 * (B,2,R,R) fp32 NCHW) (network/PoseNet.py:196-197, conv_pnp_net.py:72-83). w (45, Cout) fp32 tap-major, k = ci*9+kh*3+kw.
 * out (B,R/2,R/2,Cout) dtype. */
int gp_pnp_conv1(const float* xyz4, const float* coord2d, const float* w, void* y, int B, int R, int Cout,
                 int dtype, void* stream);

/* Plain Conv2d(3,Cout,3,s2,p1,bias=False) on the (B*HW,4) fp32 coordinate map: first MAPEncoder layer
 * when use_dcn='' (conv_pnp_net.py:258-272). w (27, Cout) fp32 tap-major. */
int gp_xyz_conv3x3_s2(const float* xyz4, const float* w, void* y, int B, int R, int Cout, int dtype,
                      void* stream);

/* SizeHead (network/pose_head.py:30-42) + mean-size residual (network/PoseNet.py:199-202):
 * feat (B,HW,C); w1 (F,C) / b1 (F) with eval BatchNorm folded in; w2 (3,F), b2 (3);
 * out size (B,3) fp32 = head + mean_size/||mean_size||; scratch: B*(F + C) floats (hidden units, pooled maxima). */
int gp_size_head(const void* feat, const float* w1, const float* b1, const float* w2, const float* b2,
                 const float* mean_size, float* size, float* scratch, int B, int HW, int C, int F, int dtype,
                 void* stream);

/* Pose tail: fc_r/fc_t/fc_z (conv_pnp_net.py:190-199), rot6d -> R (pose_utils/rot_reps.py:34-55),
 * centroid/z back-projection and allocentric -> egocentric (pose_from_pred_centroid_z.py:60-157,
 * pose_utils/utils.py:29-84) on device.  h / hz: (B,256) fp32 rows with stride ldh.
 * outputs fp32: rot6d (B,6), pred_t (B,3), rot_allo (B,9), rot_ego (B,9), trans (B,3). */
int gp_pose_tail(const float* h, const float* hz, int ldh, const float* w_r, const float* b_r,
                 const float* w_t, const float* b_t, const float* w_z, const float* b_z, const float* cam_K,
                 const float* bbox_center, const float* resize_ratio, const float* roi_wh, int wild6d,
                 int site_centroid, float* rot6d, float* pred_t, float* rot_allo, float* rot_ego,
                 float* trans, int B, void* stream);

/* MAPTransformerEncoer front end (network/attention_pnp_net.py:126-157, PatchEmbed :264-302): gather the PxP patches
 * of the (B*R*R,4) fp32 coordinate map into GEMM rows (B*(R/P)^2, P*P*3) of `dtype`, k = (ky*P+kx)*3 + c. */
int gp_patchify_xyz(const float* xyz4, void* out, int B, int R, int P, int dtype, void* stream);
/* Multi-head self-attention core of timm 0.9.6 vision_transformer.Attention for 64 tokens x head_dim 32: qkv rows
 * (B*64, 3*heads*32) laid out [q|k|v][head][32]; out (B*64, heads*32) = softmax(q k^T / sqrt(32)) v, fp32 math. */
int gp_attention64(const void* qkv, void* out, int B, int heads, int dtype, void* stream);

/* ResNet stem (network/resnet.py:137-141): Conv2d(3,64,7,s2,p3,bias=False) + eval BatchNorm (folded by the host into
 * w / b) + ReLU on the NCHW fp32 image -> (B,H/2,W/2,64) channels-last.  w: (147, 64) fp32 tap-major, k = c*49+kh*7+kw. */
int gp_resnet_stem(const float* img, const float* w, const float* b, void* out, int B, int H, int W, int dtype,
                   void* stream);
/* nn.MaxPool2d(3, stride 2, padding 1), channels-last (network/resnet.py:140). */
int gp_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream);

/* torchvision Resize(out, NEAREST) on a square fp32 mask (B,1,S,S) -> (B,1,R,R) (PoseNet.py:170,180). */
int gp_mask_resize_nearest(const float* mask, float* out, int B, int S, int R, void* stream);

/* Crop pre-processing on the device: the four cv2.warpAffine(..., INTER_NEAREST) crops of
 * evaluation/load_data_eval.py:262-288 (via tools/dataset_utils.py:101-114) for B detections in one launch.
 *   frames  (F,H,W,3) uint8, masks (NM,H,W) uint8 (non-zero = 1.0), frame_idx / mask_idx: (B) int32
 *   inv_img / inv_out: (B,6) float64 = the INVERTED 2x3 affine maps dst->src for the img_size (S) and out_res (R)
 *       crops, exactly what cv::warpAffine computes before its fixed-point walk (host: givepose_amd/preprocess.py)
 *   img_lut (3,256) fp32 = ((v/255 - mean[c]) / std[c]) evaluated in float64 like numpy, xlut (W) / ylut (H) fp32 =
 *       the normalised pixel grid of get_2d_coord_np (tools/dataset_utils.py:8-30)
 *   -> roi_img (B,3,S,S) fp32, roi_mask (B,1,S,S) fp32, roi_coord_2d (B,2,R,R) fp32; source pixel
 *       X = (rint((M1*y + M2)*1024) + 512 + rint(M0*x*1024)) >> 10 (Y alike), constant-0 border (so a border pixel
 *       of roi_img is the normalised value of 0, as in the reference). */
int gp_crop_rois(const unsigned char* frames, const unsigned char* masks, const int* frame_idx, const int* mask_idx,
                 const double* inv_img, const double* inv_out, const float* img_lut, const float* xlut, const float* ylut,
                 float* roi_img, float* roi_mask, float* roi_coord_2d, int B, int F, int NM, int H, int W, int S, int R,
                 void* stream);

/* Evaluation post-processing (evaluation/evaluate.py:116-125): pred_RT (B,4,4) fp32 =
 * [[R | t] * scale, 0 0 0 1] and pred_size (B,3) = size / max(||size||_2, 1e-12) (F.normalize).  R (B,9), t (B,3),
 * size (B,3), scale (B) fp32 on the device; scale may be null (= 1). */
int gp_pred_rt(const float* R, const float* t, const float* size, const float* scale, float* pred_rt, float* pred_size,
               int B, void* stream);

/* Per-crop pose row for the all-gather of the sharded path: out (B,15) fp32 = [R row-major 9 | t 3 | size 3]. */
int gp_pack_poses(const float* R, const float* t, const float* size, float* out, int B, void* stream);

/* ---- hipGraph capture of a launch sequence (launch-bound inner loop -> one graph launch) */
int gp_graph_begin(void* stream);
int gp_graph_end(void* stream, void** graph_exec_out);
int gp_graph_launch(void* graph_exec, void* stream);
int gp_graph_destroy(void* graph_exec);

/* ---------------------------------------------------------------------------------------------
 * Scale_net (network/scale_net.py:22-65; called before PoseNet at evaluation/evaluate.py:111-113) -- fp32, channels-last.
 * Each entry replaces the ATen calls of one torchvision mobilenet_v3_small building block (third-party, 0.15.2) with eval
 * BatchNorm folded into the weights by the host; act: 0 none, 1 ReLU, 2 Hardswish.
 *   gp_sn_stem      features[0]: Conv2d(3,16,3,s2,p1)+BN+Hardswish; img (B,3,H,W) NCHW, w (27,16) k = ci*9+kh*3+kw -> y (B,H/2,W/2,16)
 *   gp_sn_pointwise Conv2d 1x1 (+BN) (+act) (+residual); x (M,K) rows optionally scaled by the SqueezeExcitation gate se (M/HW, K)
 *   gp_sn_depthwise Conv2d kxk groups=C stride s pad k/2 (+BN) + act; w tap-major (k*k, C)
 *   gp_sn_avgpool   AdaptiveAvgPool2d(1): (B,HW,C) -> (B,C)
 *   gp_sn_se        SqueezeExcitation gate: hardsigmoid(fc2(relu(fc1(pooled)))) -> (B,C)
 *   gp_sn_head      scale_net.py:53-65: line1/ReLU/cat(one_hot)/line2/ReLU/cat(one_hot)/cat(roi_wh/100)/line3 + ||mean_size|| -> (B) */
int gp_sn_stem(const float* img, const float* w, const float* b, float* y, int B, int H, int W, void* stream);
int gp_sn_pointwise(const float* x, const float* w, const float* bias, const float* se, const float* residual, float* y, long M,
                    int N, int K, int HW, int act, void* stream);
int gp_sn_depthwise(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int KS, int stride,
                    int act, void* stream);
int gp_sn_avgpool(const float* x, float* y, int B, int HW, int C, void* stream);
int gp_sn_se(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* scale, int B, int C,
             int S, void* stream);
int gp_sn_head(const float* feat_roi, const float* feat_full, const float* one_hot, const float* roi_wh, const float* mean_size,
               const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, float* scale,
               int B, int F, int feat_dim, int cats_num, int use_hw, void* stream);

/* ---- per-launch HIP-event timing (bench.py roofline leg).  Between begin/end every gp_* launch on
 * `stream` is bracketed by hipEvents; gp_timing_report fills, per kernel class (GP_KC_*), launches,
 * total ms, algorithmic flops and bytes. */
enum gp_kernel_class {
    GP_KC_GEMM = 0, GP_KC_DCNV3 = 1, GP_KC_DWCONV_LN = 2, GP_KC_NORM = 3, GP_KC_ELEMENTWISE = 4,
    GP_KC_SMALL = 5, GP_KC_COUNT = 6
};
int gp_timing_begin(void* stream);
int gp_timing_end(void);
int gp_timing_report(int cls, long* launches, double* ms, double* flops, double* bytes);
/* The same launches grouped by kernel label (entry point + shape, e.g. "gemm v10 M16384 N2048 K512 epi1"), sorted by
 * total time: rank 0 is the most expensive.  Returns GP_ERR_INVALID past the last group. */
int gp_timing_top(int rank, char* label, int label_len, int* cls, long* launches, double* ms, double* flops, double* bytes);

#ifdef __cplusplus
}
#endif
#endif
