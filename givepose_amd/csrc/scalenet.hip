// Scale_net (network/scale_net.py:22-65) building blocks for gfx950: fp32 storage and arithmetic, channels-last.
//
// Scale_net runs once per detection in front of PoseNet (evaluation/evaluate.py:111-113): two torchvision
// mobilenet_v3_small feature extractors (0.06 GMAC per 256x256 image each) and three small Linear layers.  It is not the
// path north_star prices; the kernels are plain VALU fp32 -- channel counts are 16..576 in steps of 8,
// eval BatchNorm is folded into the convolution weights by the host (givepose_amd/scale_net.py), activations and the
// SqueezeExcitation scale are fused into the producing / consuming kernel.
//   act codes: 0 none, 1 ReLU, 2 Hardswish (x * relu6(x + 3) / 6)
#include "common.hpp"

namespace {

__device__ __forceinline__ float sn_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return v * fminf(fmaxf(v + 3.f, 0.f), 6.f) * (1.f / 6.f);
    return v;
}

// features[0]: Conv2d(3, 16, 3, stride 2, pad 1) + BN + Hardswish.  img (B,3,H,W) NCHW -> y (B,H/2,W/2,16)
__global__ __launch_bounds__(256) void sn_stem_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ b,
                                                      float* __restrict__ y, int B, int H, int W) {
    __shared__ float ws[27 * 16 + 16];
    for (int i = threadIdx.x; i < 27 * 16 + 16; i += 256) ws[i] = i < 432 ? w[i] : b[i - 432];   // w: (27, 16), k = ci*9 + kh*3 + kw
    __syncthreads();
    const int Ho = H / 2, Wo = W / 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * Ho * Wo) return;
    const int wo = (int)(idx % Wo), ho = (int)((idx / Wo) % Ho), bi = (int)(idx / ((long)Wo * Ho));
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = ws[432 + c];
    for (int ci = 0; ci < 3; ++ci)
        for (int kh = 0; kh < 3; ++kh) {
            const int hi = ho * 2 - 1 + kh;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int wi = wo * 2 - 1 + kw;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float v = img[(((long)bi * 3 + ci) * H + hi) * W + wi];
                const float* wk = ws + (ci * 9 + kh * 3 + kw) * 16;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(v, wk[c], acc[c]);
            }
        }
    float* o = y + idx * 16;
#pragma unroll
    for (int c = 0; c < 16; c += 4)
        *reinterpret_cast<f32x4*>(o + c) = f32x4{sn_act(acc[c], 2), sn_act(acc[c + 1], 2), sn_act(acc[c + 2], 2), sn_act(acc[c + 3], 2)};
}

// 1x1 convolution: y[m][n] = act( sum_k (x[m][k] * se[b(m)][k]) * w[n][k] + bias[n] ) (+ res[m][n]).
// 64 rows x 64 columns per workgroup, K in chunks of 8 through LDS, 4 x 4 outputs per thread.
__global__ __launch_bounds__(256) void sn_pointwise_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           const float* __restrict__ se, const float* __restrict__ res, float* __restrict__ y,
                                                           long M, int N, int K, int HW, int act) {
    __shared__ float xs[8][64 + 4], wsm[8][64 + 4];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;           // thread: rows ty*4.., columns tx*4..
    const long m0 = (long)blockIdx.x * 64;
    const int n0 = blockIdx.y * 64;
    float acc[4][4] = {};
    const int lr = tid >> 2, lk = (tid & 3) * 2;                          // loader: row lr (0..63), k pair lk
    for (int k0 = 0; k0 < K; k0 += 8) {
        {
            const long m = m0 + lr;
            float a0 = 0.f, a1 = 0.f, b0 = 0.f, b1 = 0.f;
            if (m < M && k0 + lk < K) {
                a0 = x[m * K + k0 + lk]; a1 = x[m * K + k0 + lk + 1];
                if (se) { const float* s = se + (m / HW) * K + k0 + lk; a0 *= s[0]; a1 *= s[1]; }
            }
            const int n = n0 + lr;
            if (n < N && k0 + lk < K) { b0 = w[(long)n * K + k0 + lk]; b1 = w[(long)n * K + k0 + lk + 1]; }
            xs[lk][lr] = a0; xs[lk + 1][lr] = a1; wsm[lk][lr] = b0; wsm[lk + 1][lr] = b1;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(&xs[k][ty * 4]);
            const f32x4 b = *reinterpret_cast<const f32x4*>(&wsm[k][tx * 4]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
    const int n = n0 + tx * 4;
    if (n >= N) return;                                                   // N % 4 == 0
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + n);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long m = m0 + ty * 4 + i;
        if (m >= M) break;
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = sn_act(acc[i][j] + bv[j], act);
        if (res) o += *reinterpret_cast<const f32x4*>(res + m * N + n);
        *reinterpret_cast<f32x4*>(y + m * N + n) = o;
    }
}

// depth-wise k x k (3 or 5), stride 1 or 2, pad k/2, + bias + act.  x (B,H,W,C) -> y (B,Ho,Wo,C); w tap-major (k*k, C)
__global__ __launch_bounds__(256) void sn_depthwise_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ y, int B, int H, int W, int C, int KS, int stride, int act) {
    const int C4 = C >> 2, Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride, pad = KS / 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * Ho * Wo * C4) return;
    const int c = (int)(idx % C4) * 4;
    long t = idx / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int bi = (int)(t / Ho);
    f32x4 acc = *reinterpret_cast<const f32x4*>(bias + c);
    for (int kh = 0; kh < KS; ++kh) {
        const int hi = ho * stride - pad + kh;
        if ((unsigned)hi >= (unsigned)H) continue;
        for (int kw = 0; kw < KS; ++kw) {
            const int wi = wo * stride - pad + kw;
            if ((unsigned)wi >= (unsigned)W) continue;
            const f32x4 v = *reinterpret_cast<const f32x4*>(x + (((long)bi * H + hi) * W + wi) * C + c);
            const f32x4 f = *reinterpret_cast<const f32x4*>(w + (kh * KS + kw) * C + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(v[j], f[j], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = sn_act(acc[j], act);
    *reinterpret_cast<f32x4*>(y + (((long)bi * Ho + ho) * Wo + wo) * C + c) = acc;
}

// global average pool (B,HW,C) -> (B,C): block = (image, 64 channels), 4 pixel lanes per channel, fixed summation order
__global__ __launch_bounds__(256) void sn_avgpool_kernel(const float* __restrict__ x, float* __restrict__ y, int HW, int C) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6, b = blockIdx.y;
    float s = 0.f;
    if (c < C)
        for (int p = pl; p < HW; p += 4) s += x[((long)b * HW + p) * C + c];
    part[pl][threadIdx.x & 63] = s;
    __syncthreads();
    if (pl == 0 && c < C) y[(long)b * C + c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) / (float)HW;
}

// SqueezeExcitation gate: scale[b][c] = hardsigmoid( fc2( relu( fc1(pooled[b]) ) ) ); one workgroup per image
__global__ __launch_bounds__(256) void sn_se_kernel(const float* __restrict__ pooled, const float* __restrict__ w1, const float* __restrict__ b1,
                                                    const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ scale, int C, int S) {
    __shared__ float p[576], h[160];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < C; i += 256) p[i] = pooled[(long)b * C + i];
    __syncthreads();
    for (int s = threadIdx.x; s < S; s += 256) {
        float a = b1[s];
        for (int k = 0; k < C; ++k) a = fmaf(w1[(long)s * C + k], p[k], a);
        h[s] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = b2[c];
        for (int k = 0; k < S; ++k) a = fmaf(w2[(long)c * S + k], h[k], a);
        scale[(long)b * C + c] = fminf(fmaxf(a + 3.f, 0.f), 6.f) * (1.f / 6.f);
    }
}

// network/scale_net.py:53-65: line1 -> ReLU -> cat(one_hot) -> line2 -> ReLU -> cat(one_hot) -> cat(roi_wh / 100) -> line3, + ||mean_size||
__global__ __launch_bounds__(256) void sn_head_kernel(const float* __restrict__ f_roi, const float* __restrict__ f_full, const float* __restrict__ one_hot,
                                                      const float* __restrict__ roi_wh, const float* __restrict__ mean_size,
                                                      const float* __restrict__ w1, const float* __restrict__ b1, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, const float* __restrict__ w3, const float* __restrict__ b3,
                                                      float* __restrict__ out, int F, int FD, int NC, int use_hw) {
    __shared__ float feat[1152], x1[128 + 16], x2[64 + 16 + 2];
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < 2 * F; i += 256) feat[i] = i < F ? f_roi[(long)b * F + i] : f_full[(long)b * F + i - F];
    __syncthreads();
    if (tid < 128) {
        float a = b1[tid];
        for (int k = 0; k < 2 * F; ++k) a = fmaf(w1[(long)tid * 2 * F + k], feat[k], a);
        x1[tid] = fmaxf(a, 0.f);
    } else if (tid < 128 + NC) {
        x1[tid] = one_hot[(long)b * NC + tid - 128];
    }
    __syncthreads();
    if (tid < FD) {
        float a = b2[tid];
        for (int k = 0; k < 128 + NC; ++k) a = fmaf(w2[(long)tid * (128 + NC) + k], x1[k], a);
        x2[tid] = fmaxf(a, 0.f);
    } else if (tid < FD + NC) {
        x2[tid] = one_hot[(long)b * NC + tid - FD];
    } else if (use_hw && tid < FD + NC + 2) {
        x2[tid] = roi_wh[(long)b * 2 + tid - FD - NC] / 100.f;
    }
    __syncthreads();
    if (tid == 0) {
        const int n3 = FD + NC + (use_hw ? 2 : 0);
        float a = b3[0];
        for (int k = 0; k < n3; ++k) a = fmaf(w3[k], x2[k], a);
        const float* ms = mean_size + (long)b * 3;
        out[b] = a + sqrtf(ms[0] * ms[0] + ms[1] * ms[1] + ms[2] * ms[2]);
    }
}

}  // namespace

extern "C" int gp_sn_stem(const float* img, const float* w, const float* b, float* y, int B, int H, int W, void* stream) {
    GP_REQUIRE(img && w && b && y && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "gp_sn_stem: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long px = (long)B * (H / 2) * (W / 2);
    gp_timing_before(s, GP_KC_SMALL, 2.0 * px * 27 * 16, (double)B * 3 * H * W * 4 + px * 64.0);
    hipLaunchKernelGGL(sn_stem_kernel, dim3((unsigned)cdiv(px, 256)), dim3(256), 0, s, img, w, b, y, B, H, W);
    GP_LAUNCH_CHECK("gp_sn_stem");
}

extern "C" int gp_sn_pointwise(const float* x, const float* w, const float* bias, const float* se, const float* residual, float* y, long M,
                               int N, int K, int HW, int act, void* stream) {
    GP_REQUIRE(x && w && bias && y && M > 0 && N > 0 && K > 0 && HW > 0, "gp_sn_pointwise: bad argument");
    GP_REQUIRE(N % 4 == 0 && K % 2 == 0 && M % HW == 0 && act >= 0 && act <= 2, "gp_sn_pointwise: N=%d %% 4, K=%d %% 2, M %% HW required", N, K);
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 2.0 * M * N * K, ((double)M * (K + N) + (double)N * K) * 4);
    gp_timing_label("sn_pointwise M%ld N%d K%d", M, N, K);
    hipLaunchKernelGGL(sn_pointwise_kernel, dim3((unsigned)cdiv(M, 64), (unsigned)cdiv(N, 64)), dim3(256), 0, s, x, w, bias, se, residual, y, M, N, K, HW, act);
    GP_LAUNCH_CHECK("gp_sn_pointwise");
}

extern "C" int gp_sn_depthwise(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int KS, int stride,
                               int act, void* stream) {
    GP_REQUIRE(x && w && bias && y && B > 0 && H > 0 && W > 0, "gp_sn_depthwise: bad argument");
    GP_REQUIRE(C % 4 == 0 && (KS == 3 || KS == 5) && (stride == 1 || stride == 2) && act >= 0 && act <= 2, "gp_sn_depthwise: C=%d KS=%d stride=%d", C, KS, stride);
    hipStream_t s = (hipStream_t)stream;
    const int Ho = (H + stride - 1) / stride, Wo = (W + stride - 1) / stride;
    const long n = (long)B * Ho * Wo * (C / 4);
    gp_timing_before(s, GP_KC_SMALL, 2.0 * n * 4 * KS * KS, ((double)B * H * W * C + (double)n * 4) * 4);
    hipLaunchKernelGGL(sn_depthwise_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, s, x, w, bias, y, B, H, W, C, KS, stride, act);
    GP_LAUNCH_CHECK("gp_sn_depthwise");
}

extern "C" int gp_sn_avgpool(const float* x, float* y, int B, int HW, int C, void* stream) {
    GP_REQUIRE(x && y && B > 0 && HW > 0 && C > 0, "gp_sn_avgpool: bad argument");
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, (double)B * HW * C, (double)B * HW * C * 4);
    hipLaunchKernelGGL(sn_avgpool_kernel, dim3((unsigned)cdiv(C, 64), (unsigned)B), dim3(256), 0, s, x, y, HW, C);
    GP_LAUNCH_CHECK("gp_sn_avgpool");
}

extern "C" int gp_sn_se(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* scale, int B, int C,
                        int S, void* stream) {
    GP_REQUIRE(pooled && w1 && b1 && w2 && b2 && scale && B > 0, "gp_sn_se: bad argument");
    GP_REQUIRE(C > 0 && C <= 576 && S > 0 && S <= 160, "gp_sn_se: C=%d (<= 576), S=%d (<= 160)", C, S);
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 4.0 * B * C * S, (double)2 * C * S * 4);
    hipLaunchKernelGGL(sn_se_kernel, dim3((unsigned)B), dim3(256), 0, s, pooled, w1, b1, w2, b2, scale, C, S);
    GP_LAUNCH_CHECK("gp_sn_se");
}

extern "C" int gp_sn_head(const float* feat_roi, const float* feat_full, const float* one_hot, const float* roi_wh, const float* mean_size,
                          const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, float* scale,
                          int B, int F, int feat_dim, int cats_num, int use_hw, void* stream) {
    GP_REQUIRE(feat_roi && feat_full && one_hot && roi_wh && mean_size && w1 && b1 && w2 && b2 && w3 && b3 && scale && B > 0, "gp_sn_head: bad argument");
    GP_REQUIRE(F == 576 && feat_dim > 0 && feat_dim <= 64 && cats_num > 0 && cats_num <= 16, "gp_sn_head: F=%d feat_dim=%d cats_num=%d", F, feat_dim, cats_num);
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 2.0 * B * (2 * F * 128 + 134 * feat_dim), (double)2 * F * 128 * 4);
    hipLaunchKernelGGL(sn_head_kernel, dim3((unsigned)B), dim3(256), 0, s, feat_roi, feat_full, one_hot, roi_wh, mean_size, w1, b1, w2, b2, w3, b3,
                       scale, F, feat_dim, cats_num, use_hw);
    GP_LAUNCH_CHECK("gp_sn_head");
}
