/* ORACLE -- test infrastructure, not product code.
 *
 * Plain-C scalar restatement of the reference's DCNv3 forward CUDA kernel
 *   network/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:216-282  (dcnv3_im2col_gpu_kernel)
 *   network/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:32-80    (dcnv3_im2col_bilinear)
 *   network/ops_dcnv3/src/cuda/dcnv3_cuda.cu:21-85            (host wrapper: output geometry,
 *                                                              flat offset/mask addressing)
 * one loop iteration per CUDA thread, fp32 storage, fp32 accumulation (opmath_t of float).
 * Pinned by tests/golden/dcnv3_s1.npz (the reference test's own parameters, expected output from
 * the reference's dcnv3_core_pytorch) and dcnv3_s2_B{1,4,5}.npz (tests/test_oracle_golden.py).
 * Build: make -C oracle   (gcc -O2 -shared -fPIC)
 */
#include <math.h>
#include <stddef.h>

static float bilinear(const float *im, int H, int W, int G, int D, float h, float w, int g, int c) {
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
    const int w_stride = G * D, h_stride = W * w_stride;
    const int base = g * D + c;
    float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = im[h_low * h_stride + w_low * w_stride + base];
    if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * h_stride + w_high * w_stride + base];
    if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * h_stride + w_low * w_stride + base];
    if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * h_stride + w_high * w_stride + base];
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

/* in: (N,H,W,G*D); offset/mask: FLAT buffers, consumed from index 0 by ((b*Ho+ho)*Wo+wo)*G+g;
 * out: (N,Ho,Wo,G*D). Returns 0, or -1 on bad geometry. */
int dcnv3_forward_c(const float *in, const float *offset, const float *mask, float *out, int N, int H,
                    int W, int G, int D, int K, int stride, int pad, int dil, float offset_scale,
                    int remove_center) {
    const int Ho = (H + 2 * pad - (dil * (K - 1) + 1)) / stride + 1;
    const int Wo = (W + 2 * pad - (dil * (K - 1) + 1)) / stride + 1;
    if (Ho <= 0 || Wo <= 0 || G <= 0 || D <= 0) return -1;
    const int P = K * K - remove_center;
    const long total = (long)N * Ho * Wo * G * D;
    for (long index = 0; index < total; ++index) {
        long t = index;
        const int c = (int)(t % D); t /= D;
        const long sampling_index = t;
        const int g = (int)(t % G); t /= G;
        const int p0_w = ((dil * (K - 1)) >> 1) - pad + (int)(t % Wo) * stride; t /= Wo;
        const int p0_h = ((dil * (K - 1)) >> 1) - pad + (int)(t % Ho) * stride; t /= Ho;
        const int b = (int)t;
        long wptr = sampling_index * P, lptr = wptr << 1;
        const float *im = in + (size_t)b * H * W * G * D;
        const float p0_w_ = p0_w - ((dil * (K - 1)) >> 1) * offset_scale;
        const float p0_h_ = p0_h - ((dil * (K - 1)) >> 1) * offset_scale;
        float col = 0;
        for (int i = 0; i < K; ++i)
            for (int j = 0; j < K; ++j) {
                if (remove_center && i == K / 2 && j == K / 2) continue;
                const float ow = offset[lptr], oh = offset[lptr + 1];
                const float loc_w = p0_w_ + (i * dil + ow) * offset_scale;
                const float loc_h = p0_h_ + (j * dil + oh) * offset_scale;
                const float wgt = mask[wptr];
                if (loc_h > -1 && loc_w > -1 && loc_h < H && loc_w < W)
                    col += bilinear(im, H, W, G, D, loc_h, loc_w, g, c) * wgt;
                wptr += 1; lptr += 2;
            }
        out[index] = col;
    }
    return 0;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Generic geometry (independent h / w kernel, stride, pad, dilation; any D), double precision, and the BACKWARD:
 *   forward : dcnv3_im2col_cuda.cuh:216-282 with opmath_t = double (dcnv3_cuda.cu:68 dispatches double)
 *   backward: dcnv3_im2col_cuda.cuh:386-487 (dcnv3_col2im_gpu_kernel_shm_blocksize_aware_reduce_v2) +
 *             :82-140 (dcnv3_col2im_bilinear); host dcnv3_cuda.cu:87-174 -- grad buffers are zero-filled, grad_input is
 *             accumulated over every (pixel, tap, channel), grad_offset / grad_mask are the sums over the D channels of
 *             a group.  The summation order over channels is sequential here (the CUDA kernel uses a shared-memory tree).
 * Pinned by tests/golden/dcnv3_any_*.npz = the reference's dcnv3_core_pytorch and its autograd (scripts/gen_golden_dcnv3_any.py).
 */
typedef struct { int kh, kw, sh, sw, ph, pw, dh, dw, G, D, rc; } dcn_geom;

static double bilinear_d(const double *im, int H, int W, int G, int D, double h, double w, int g, int c) {
    const int h_low = (int)floor(h), w_low = (int)floor(w);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const double lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
    const int w_stride = G * D, h_stride = W * w_stride, base = g * D + c;
    double v1 = 0, v2 = 0, v3 = 0, v4 = 0;
    if (h_low >= 0 && w_low >= 0) v1 = im[h_low * h_stride + w_low * w_stride + base];
    if (h_low >= 0 && w_high <= W - 1) v2 = im[h_low * h_stride + w_high * w_stride + base];
    if (h_high <= H - 1 && w_low >= 0) v3 = im[h_high * h_stride + w_low * w_stride + base];
    if (h_high <= H - 1 && w_high <= W - 1) v4 = im[h_high * h_stride + w_high * w_stride + base];
    return hh * hw * v1 + hh * lw * v2 + lh * hw * v3 + lh * lw * v4;
}

int dcnv3_forward_any_c(const double *in, const double *offset, const double *mask, double *out, int N, int H, int W,
                        const dcn_geom *q, double offset_scale) {
    const int Ho = (H + 2 * q->ph - (q->dh * (q->kh - 1) + 1)) / q->sh + 1;
    const int Wo = (W + 2 * q->pw - (q->dw * (q->kw - 1) + 1)) / q->sw + 1;
    if (Ho <= 0 || Wo <= 0) return -1;
    const int G = q->G, D = q->D, P = q->kh * q->kw - q->rc;
    const long total = (long)N * Ho * Wo * G * D;
    for (long index = 0; index < total; ++index) {
        long t = index;
        const int c = (int)(t % D); t /= D;
        const long sampling_index = t;
        const int g = (int)(t % G); t /= G;
        const int p0_w = ((q->dw * (q->kw - 1)) >> 1) - q->pw + (int)(t % Wo) * q->sw; t /= Wo;
        const int p0_h = ((q->dh * (q->kh - 1)) >> 1) - q->ph + (int)(t % Ho) * q->sh; t /= Ho;
        const int b = (int)t;
        long wptr = sampling_index * P, lptr = wptr << 1;
        const double *im = in + (size_t)b * H * W * G * D;
        const double p0_w_ = p0_w - ((q->dw * (q->kw - 1)) >> 1) * offset_scale;
        const double p0_h_ = p0_h - ((q->dh * (q->kh - 1)) >> 1) * offset_scale;
        double col = 0;
        for (int i = 0; i < q->kw; ++i)
            for (int j = 0; j < q->kh; ++j) {
                if (q->rc && i == q->kw / 2 && j == q->kh / 2) continue;
                const double loc_w = p0_w_ + (i * q->dw + offset[lptr]) * offset_scale;
                const double loc_h = p0_h_ + (j * q->dh + offset[lptr + 1]) * offset_scale;
                if (loc_h > -1 && loc_w > -1 && loc_h < H && loc_w < W)
                    col += bilinear_d(im, H, W, G, D, loc_h, loc_w, g, c) * mask[wptr];
                wptr += 1; lptr += 2;
            }
        out[index] = col;
    }
    return 0;
}

/* grad_in (N,H,W,G*D), grad_offset / grad_mask: flat, same extent as the consumed offset / mask prefix; all three must
 * be zero-filled by the caller (dcnv3_cuda.cu:128-130). */
int dcnv3_backward_any_c(const double *in, const double *offset, const double *mask, const double *grad_out,
                         double *grad_in, double *grad_offset, double *grad_mask, int N, int H, int W,
                         const dcn_geom *q, double offset_scale) {
    const int Ho = (H + 2 * q->ph - (q->dh * (q->kh - 1) + 1)) / q->sh + 1;
    const int Wo = (W + 2 * q->pw - (q->dw * (q->kw - 1) + 1)) / q->sw + 1;
    if (Ho <= 0 || Wo <= 0) return -1;
    const int G = q->G, D = q->D, P = q->kh * q->kw - q->rc;
    const int w_stride = G * D, h_stride = W * w_stride;
    const long total = (long)N * Ho * Wo * G * D;
    for (long index = 0; index < total; ++index) {
        long t = index;
        const int c = (int)(t % D); t /= D;
        const long sampling_index = t;
        const int g = (int)(t % G); t /= G;
        const int p0_w = ((q->dw * (q->kw - 1)) >> 1) - q->pw + (int)(t % Wo) * q->sw; t /= Wo;
        const int p0_h = ((q->dh * (q->kh - 1)) >> 1) - q->ph + (int)(t % Ho) * q->sh; t /= Ho;
        const int b = (int)t;
        const double top_grad = grad_out[index];
        long wptr = sampling_index * P, lptr = wptr << 1;
        const double *im = in + (size_t)b * H * W * G * D;
        double *gim = grad_in + (size_t)b * H * W * G * D;
        const double p0_w_ = p0_w - ((q->dw * (q->kw - 1)) >> 1) * offset_scale;
        const double p0_h_ = p0_h - ((q->dh * (q->kh - 1)) >> 1) * offset_scale;
        const int base = g * D + c;
        for (int i = 0; i < q->kw; ++i)
            for (int j = 0; j < q->kh; ++j) {
                if (q->rc && i == q->kw / 2 && j == q->kh / 2) continue;
                const double w = p0_w_ + (i * q->dw + offset[lptr]) * offset_scale;
                const double h = p0_h_ + (j * q->dh + offset[lptr + 1]) * offset_scale;
                const double m = mask[wptr];
                if (h > -1 && w > -1 && h < H && w < W) {      /* dcnv3_col2im_bilinear, :82-140 */
                    const int h_low = (int)floor(h), w_low = (int)floor(w), h_high = h_low + 1, w_high = w_low + 1;
                    const double lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
                    const double w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw, tg = top_grad * m;
                    double gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                    if (h_low >= 0 && w_low >= 0) { const int p = h_low * h_stride + w_low * w_stride + base; v1 = im[p]; gh -= hw * v1; gw -= hh * v1; gim[p] += w1 * tg; }
                    if (h_low >= 0 && w_high <= W - 1) { const int p = h_low * h_stride + w_high * w_stride + base; v2 = im[p]; gh -= lw * v2; gw += hh * v2; gim[p] += w2 * tg; }
                    if (h_high <= H - 1 && w_low >= 0) { const int p = h_high * h_stride + w_low * w_stride + base; v3 = im[p]; gh += hw * v3; gw -= lh * v3; gim[p] += w3 * tg; }
                    if (h_high <= H - 1 && w_high <= W - 1) { const int p = h_high * h_stride + w_high * w_stride + base; v4 = im[p]; gh += lw * v4; gw += lh * v4; gim[p] += w4 * tg; }
                    grad_mask[wptr] += top_grad * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                    grad_offset[lptr] += offset_scale * gw * tg;
                    grad_offset[lptr + 1] += offset_scale * gh * tg;
                }
                wptr += 1; lptr += 2;
            }
    }
    return 0;
}
