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
