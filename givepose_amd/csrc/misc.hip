// Small / bandwidth-bound kernels of the PoseNet path (gfx950): ConvNeXt stem, bilinear x2,
// deconv col2im, xyz out layer, tiny-Cin convolutions, SizeHead, pose tail, mask resize.
#include "common.hpp"

namespace {

__device__ __forceinline__ void store_T(half_t* p, float v) { *p = (half_t)v; }
__device__ __forceinline__ void store_T(float* p, float v) { *p = v; }

// ------------------------------------------------------------------------------------- stem
// conv4x4 s4 (3 -> 128) + LayerNorm over the 128 channels.  Block = PXB output pixels of one row; a wave owns
// PXB/4 pixels and its 64 lanes own the 64 channel pairs of a pixel, so the LayerNorm reduction is a pure
// wave butterfly.  The 48 x 2 filter taps of a lane's channel pair live in registers; the input patch rows are
// staged once in LDS and read back as broadcast 16-B vectors (one (c,kh) row of 4 taps per read).
template <typename T>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, const float* __restrict__ wt,
                                                   const float* __restrict__ bias, const float* __restrict__ lnw,
                                                   const float* __restrict__ lnb, T* __restrict__ out, int H,
                                                   int W, float eps, int PXB) {
    constexpr int C0 = 128;
    __shared__ __attribute__((aligned(16))) float in_s[12][256];
    const int Ho = H / 4, Wo = W / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nwb = Wo / PXB;
    const int wblk = blockIdx.x % nwb;
    const int ho = (blockIdx.x / nwb) % Ho;
    const int b = blockIdx.x / (nwb * Ho);
    const int wo0 = wblk * PXB;
    for (int i = tid; i < 12 * PXB * 4; i += 256) {
        const int row = i / (PXB * 4), col = i - row * (PXB * 4);
        const int c = row >> 2, kh = row & 3;
        in_s[row][col] = img[(((long)b * 3 + c) * H + (ho * 4 + kh)) * W + wo0 * 4 + col];
    }
    float2 wr[48];
#pragma unroll
    for (int k = 0; k < 48; ++k) wr[k] = *reinterpret_cast<const float2*>(wt + k * C0 + lane * 2);
    const float2 bv = *reinterpret_cast<const float2*>(bias + lane * 2);
    const float2 gw = *reinterpret_cast<const float2*>(lnw + lane * 2), gb = *reinterpret_cast<const float2*>(lnb + lane * 2);
    __syncthreads();
    const int ppw = PXB / 4;
    for (int q = 0; q < ppw; ++q) {
        const int p = wave * ppw + q;
        f32x2 a01 = {bv.x, bv.y};     // both channels of the lane side by side (two v_fma_f32 per tap: no packed fp32 ops in this build)
#pragma unroll
        for (int row = 0; row < 12; ++row) {
            const f32x4 iv = *reinterpret_cast<const f32x4*>(&in_s[row][p * 4]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                a01 = __builtin_elementwise_fma(f32x2{iv[j], iv[j]}, f32x2{wr[row * 4 + j].x, wr[row * 4 + j].y}, a01);
        }
        float a0 = a01[0], a1 = a01[1];
        const float mean = group_sum(a0 + a1, 64) * (1.0f / C0);
        a0 -= mean;
        a1 -= mean;
        const float rstd = rsqrtf(group_sum(a0 * a0 + a1 * a1, 64) * (1.0f / C0) + eps);
        T* o = out + (((long)b * Ho + ho) * Wo + wo0 + p) * C0 + lane * 2;
        const float o0 = a0 * rstd * gw.x + gb.x, o1 = a1 * rstd * gw.y + gb.y;
        if constexpr (sizeof(T) == 2) *reinterpret_cast<half2v*>(o) = half2v{(half_t)o0, (half_t)o1};   // one 4-byte store
        else *reinterpret_cast<f32x2*>(o) = f32x2{o0, o1};
    }
}


// The same stem on the matrix pipe (fp16 output, 64 output pixels of a row per workgroup): as a GEMM it is M = pixels, N = 128,
// K = 48, and the VALU form above spends 48 two-channel FMA pairs per pixel and lane (45 us of the 87 at bs 64).  The fp32 image and
// the fp32 taps are both split into fp16 hi + lo, and x_hi w_hi + x_lo w_hi + x_hi w_lo is accumulated in fp32 (the dropped
// x_lo w_lo is 2^-22 of the product): the stem stays an fp32-accurate convolution, unlike the rest of the fp16 mode whose
// weights are rounded once.  K order: k = c*16 + kh*4 + kw as in wt, padded to 64 (two 32-deep MFMA steps).
// Wave w owns channels [32 w, 32 w + 32) (two 16-row A tiles, taps converted once per workgroup from wt) for all four
// 16-pixel tiles; the B fragments (pixel = column) are built from the staged patch rows: the 8 taps of a lane are two
// consecutive (c, kh) rows x 4 kw = two 16-byte LDS reads.  LayerNorm: two-pass like the VALU form, channel sums across
// the four waves through LDS.
template <int RPW>   // output rows per workgroup: the tap conversion above is paid once per RPW x 64 pixels
__global__ __launch_bounds__(256) void stem_mfma_kernel(const float* __restrict__ img, const float* __restrict__ wt,
                                                        const float* __restrict__ bias, const float* __restrict__ lnw,
                                                        const float* __restrict__ lnb, half_t* __restrict__ out, int H,
                                                        int W, float eps) {
    constexpr int C0 = 128, PXB = 64;
    __shared__ __attribute__((aligned(16))) float in_s[13][256];   // row 12: zeros (the K padding reads it)
    __shared__ float red_s[4][PXB];
    __shared__ float stat_s[PXB];
    const int Ho = H / 4, Wo = W / 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int nwb = Wo / PXB;
    const int wblk = blockIdx.x % nwb;
    const int hog = (blockIdx.x / nwb) % (Ho / RPW);
    const int b = blockIdx.x / (nwb * (Ho / RPW));
    const int wo0 = wblk * PXB;
    in_s[12][tid] = 0.f;
    // ---- A fragments: lane (fr, fq) of (nt, s) holds taps k = s*32 + fq*8 + j of channel n = 32 wave + 16 nt + fr
    half8 ah[2][2], al[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = st * 32 + fq * 8 + j;
                const float w = k < 48 ? wt[k * C0 + wave * 32 + nt * 16 + fr] : 0.f;
                const half_t hi = (half_t)w;
                ah[nt][st][j] = hi;
                al[nt][st][j] = (half_t)(w - (float)hi);
            }
    f32x4 bv[2], gw[2], gb[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = wave * 32 + nt * 16 + fq * 4;
        bv[nt] = *reinterpret_cast<const f32x4*>(bias + n);
        gw[nt] = *reinterpret_cast<const f32x4*>(lnw + n);
        gb[nt] = *reinterpret_cast<const f32x4*>(lnb + n);
    }
    // the patch rows of output row hr + 1 are fetched into registers while row hr is multiplied and normalised (three 16-byte pieces per thread):
    // a workgroup's rows used to be load -> wait -> compute one after the other
    f32x4 nxt[3];
    auto fetch_row = [&](int hr) {
        const int ho = hog * RPW + hr;
#pragma unroll
        for (int k = 0; k < 3; ++k) {              // 12 (c, kh) rows x 64 16-byte pieces
            const int i = tid + 256 * k, row = i >> 6, c4 = i & 63;
            const int c = row >> 2, kh = row & 3;
            nxt[k] = *reinterpret_cast<const f32x4*>(img + (((long)b * 3 + c) * H + (ho * 4 + kh)) * W + wo0 * 4 + c4 * 4);
        }
    };
    fetch_row(0);
    for (int hr = 0; hr < RPW; ++hr) {
    const int ho = hog * RPW + hr;
    if (hr) __syncthreads();                       // every wave is done with the previous row's patch and statistics
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int i = tid + 256 * k;
        *reinterpret_cast<f32x4*>(&in_s[i >> 6][(i & 63) * 4]) = nxt[k];
    }
    __syncthreads();
    if (hr + 1 < RPW) fetch_row(hr + 1);
    f32x4 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        // B fragments of pixel p = mt*16 + fr: taps k0 .. k0+7 with k0 = st*32 + fq*8 = rows (k0 >> 2), (k0 >> 2) + 1 of in_s
        half8 bh[2], bl[2];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int r0 = min(st * 8 + fq * 2, 12), r1 = min(st * 8 + fq * 2 + 1, 12);
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(&in_s[r0][(mt * 16 + fr) * 4]);
            const f32x4 x1 = *reinterpret_cast<const f32x4*>(&in_s[r1][(mt * 16 + fr) * 4]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = j < 4 ? x0[j & 3] : x1[j & 3];
                const half_t hi = (half_t)x;
                bh[st][j] = hi;
                bl[st][j] = (half_t)(x - (float)hi);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x4 a = bv[nt];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[nt][st], bh[st], a, 0, 0, 0);   // small terms first
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nt][st], bl[st], a, 0, 0, 0);
            }
#pragma unroll
            for (int st = 0; st < 2; ++st) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nt][st], bh[st], a, 0, 0, 0);
            acc[mt][nt] = a;
        }
    }
    // ---- LayerNorm over the 128 channels of a pixel: lane holds channels 32 wave + 16 nt + 4 fq + e of pixels mt*16 + fr
    float mean[4], rstd[4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            float a = 0.f;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (pass == 1) acc[mt][nt][e] -= mean[mt];
                    a += pass == 0 ? acc[mt][nt][e] : acc[mt][nt][e] * acc[mt][nt][e];
                }
            a += __shfl_xor(a, 16);
            a += __shfl_xor(a, 32);
            if (fq == 0) red_s[wave][mt * 16 + fr] = a;
        }
        __syncthreads();
        if (tid < PXB) {
            const float a = (red_s[0][tid] + red_s[1][tid]) + (red_s[2][tid] + red_s[3][tid]);
            stat_s[tid] = pass == 0 ? a * (1.0f / C0) : rsqrtf(a * (1.0f / C0) + eps);
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            if (pass == 0) mean[mt] = stat_s[mt * 16 + fr];
            else rstd[mt] = stat_s[mt * 16 + fr];
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            half4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)(acc[mt][nt][e] * rstd[mt] * gw[nt][e] + gb[nt][e]);
            *reinterpret_cast<half4*>(out + (((long)b * Ho + ho) * Wo + wo0 + mt * 16 + fr) * C0 + wave * 32 + nt * 16 + fq * 4) = o;
        }
    }
}

// ------------------------------------------------------------------------------------- crop pre-processing
// One thread per destination pixel of the S x S crop (image + mask) or of the R x R crop (coordinate grid); the
// fixed-point source coordinate is cv::warpAffine's (AB_BITS = 10, round-half-even of the double products).
__device__ __forceinline__ bool warp_src(const double* __restrict__ m, int x, int y, int W, int H, int& X, int& Y) {
    const int X0 = __double2int_rn((m[1] * y + m[2]) * 1024.0) + 512, Y0 = __double2int_rn((m[4] * y + m[5]) * 1024.0) + 512;
    X = (X0 + __double2int_rn(m[0] * x * 1024.0)) >> 10;
    Y = (Y0 + __double2int_rn(m[3] * x * 1024.0)) >> 10;
    return (unsigned)X < (unsigned)W && (unsigned)Y < (unsigned)H;
}

__global__ __launch_bounds__(256) void crop_rois_kernel(const unsigned char* __restrict__ frames,
                                                        const unsigned char* __restrict__ masks,
                                                        const int* __restrict__ frame_idx, const int* __restrict__ mask_idx,
                                                        const double* __restrict__ inv_img, const double* __restrict__ inv_out,
                                                        const float* __restrict__ img_lut, const float* __restrict__ xlut,
                                                        const float* __restrict__ ylut, float* __restrict__ roi_img,
                                                        float* __restrict__ roi_mask, float* __restrict__ roi_coord,
                                                        int H, int W, int S, int R) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    int X, Y;
    if (i < S * S) {
        const int y = i / S, x = i - y * S;
        const bool ok = warp_src(inv_img + b * 6, x, y, W, H, X, Y);
        const unsigned char* px = frames + ((long)frame_idx[b] * H * W + (ok ? (long)Y * W + X : 0)) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) roi_img[((long)b * 3 + c) * S * S + i] = img_lut[c * 256 + (ok ? px[c] : 0)];
        roi_mask[(long)b * S * S + i] = (ok && masks[((long)mask_idx[b] * H + Y) * W + X]) ? 1.0f : 0.0f;
    }
    if (i < R * R) {
        const int y = i / R, x = i - y * R;
        const bool ok = warp_src(inv_out + b * 6, x, y, W, H, X, Y);
        roi_coord[((long)b * 2 + 0) * R * R + i] = ok ? xlut[X] : 0.0f;
        roi_coord[((long)b * 2 + 1) * R * R + i] = ok ? ylut[Y] : 0.0f;
    }
}


// ------------------------------------------------------------------------------------- eval post-processing
__global__ __launch_bounds__(64) void pred_rt_kernel(const float* __restrict__ R, const float* __restrict__ t,
                                                     const float* __restrict__ size, const float* __restrict__ scale,
                                                     float* __restrict__ rt, float* __restrict__ ps, int B) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float sc = scale ? scale[b] : 1.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) rt[b * 16 + i * 4 + j] = R[b * 9 + i * 3 + j] * sc;
        rt[b * 16 + i * 4 + 3] = t[b * 3 + i] * sc;
    }
    rt[b * 16 + 12] = 0.f; rt[b * 16 + 13] = 0.f; rt[b * 16 + 14] = 0.f; rt[b * 16 + 15] = 1.f;
    const float x = size[b * 3], y = size[b * 3 + 1], z = size[b * 3 + 2];
    const float n = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
    ps[b * 3] = x / n; ps[b * 3 + 1] = y / n; ps[b * 3 + 2] = z / n;
}

// ------------------------------------------------------------------------------------- bilinear x2
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H,
                                                         int W, int C) {
    constexpr int VEC = Vec16<T>::N;
    const int CT = C / VEC, Ho = 2 * H, Wo = 2 * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * Ho * Wo * CT) return;
    const int cs = (int)(idx % CT);
    long t = idx / CT;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const long b = t / Ho;
    // align_corners=True: src = dst * (in-1)/(out-1)
    float sy, sx;
    if constexpr (sizeof(T) == 2) {   // fp16 storage: roundings pinned (common.hpp bilerp_*: gp_groupnorm_upsample2x must agree bit for bit)
        sy = bilerp_src((float)(H - 1) / (float)(Ho - 1), oy);
        sx = bilerp_src((float)(W - 1) / (float)(Wo - 1), ox);
    } else {                          // fp32 storage: the plain expressions of rounds 1-3
        sy = (float)(H - 1) / (float)(Ho - 1) * oy;
        sx = (float)(W - 1) / (float)(Wo - 1) * ox;
    }
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const T* xb = x + (b * H * W) * C + cs * VEC;
    const Vec16<T> v00 = load16<T>(xb + ((long)y0 * W + x0) * C), v01 = load16<T>(xb + ((long)y0 * W + x1) * C),
                   v10 = load16<T>(xb + ((long)y1 * W + x0) * C), v11 = load16<T>(xb + ((long)y1 * W + x1) * C);
    Vec16<T> o;
    if constexpr (sizeof(T) == 2) {
        float h0[VEC], h1[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            h0[e] = bilerp_h(hx, v00.get(e), lx, v01.get(e));
            h1[e] = bilerp_h(hx, v10.get(e), lx, v11.get(e));
        }
        bilerp_v_vec<T>(o, hy, h0, ly, h1);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            o.set(e, hy * (hx * v00.get(e) + lx * v01.get(e)) + ly * (hx * v10.get(e) + lx * v11.get(e)));
    }
    store16<T>(y + ((b * Ho + oy) * Wo + ox) * C + cs * VEC, o);
}

// Row-mapped form (C / VEC a power of two): blockIdx.y = output row, blockIdx.z = image, so the vertical taps are
// wave-uniform and no 64-bit div / mod is left per thread -- the flat form above spends ~2/3 of its VALU time on
// index arithmetic and is VALU bound at 2.9 TB/s (scripts/gn_bench.py).
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_rows_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W,
                                                              int C, int ct_shift, long pl) {
    constexpr int VEC = Vec16<T>::N;
    const int Ho = 2 * H, Wo = 2 * W, CT = 1 << ct_shift;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int ox = i >> ct_shift, cs = i & (CT - 1);
    if (ox >= Wo) return;
    const int oy = blockIdx.y, b = blockIdx.z;
    float sy, sx;
    if constexpr (sizeof(T) == 2) {   // fp16 storage: roundings pinned (common.hpp bilerp_*: gp_groupnorm_upsample2x must agree bit for bit)
        sy = bilerp_src((float)(H - 1) / (float)(Ho - 1), oy);
        sx = bilerp_src((float)(W - 1) / (float)(Wo - 1), ox);
    } else {                          // fp32 storage: the plain expressions of rounds 1-3
        sy = (float)(H - 1) / (float)(Ho - 1) * oy;
        sx = (float)(W - 1) / (float)(Wo - 1) * ox;
    }
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = sy - y0, lx = sx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const T* r0 = x + ((long)(b * H + y0) * W) * C + cs * VEC;
    const T* r1 = x + ((long)(b * H + y1) * W) * C + cs * VEC;
    const Vec16<T> v00 = load16<T>(r0 + x0 * C), v01 = load16<T>(r0 + x1 * C), v10 = load16<T>(r1 + x0 * C),
                   v11 = load16<T>(r1 + x1 * C);
    Vec16<T> o;
    if constexpr (sizeof(T) == 2) {
        float h0[VEC], h1[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            h0[e] = bilerp_h(hx, v00.get(e), lx, v01.get(e));
            h1[e] = bilerp_h(hx, v10.get(e), lx, v11.get(e));
        }
        bilerp_v_vec<T>(o, hy, h0, ly, h1);
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            o.set(e, hy * (hx * v00.get(e) + lx * v01.get(e)) + ly * (hx * v10.get(e) + lx * v11.get(e)));
    }
    store16p<T>(y, (((long)(b * Ho + oy)) * Wo + ox) * C + cs * VEC, o, pl);
}

// ------------------------------------------------------------------------------------- deconv col2im
// out[b][y][x][co] = sum over (ky,kx) with y = 2i-1+ky, x = 2j-1+kx of cols[(b,i,j)][(ky*3+kx)*C + co]
// CT = element type of cols: float (the fp32 / split modes), or half_t (round 5, fp16 mode: the deconv GEMM then runs its lean fp16 epilogue and
// writes half the bytes -- 38 instead of 75 MB per head at 128 crops; each of the up to four summands is rounded to fp16 once, the sum is fp32)
template <typename T, typename CT>
__global__ __launch_bounds__(256) void col2im_kernel(const CT* __restrict__ cols, T* __restrict__ out, int B,
                                                     int H, int W, int C) {
    constexpr int VEC = 16 / sizeof(CT);
    const int CV = C / VEC, Ho = 2 * H, Wo = 2 * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * Ho * Wo * CV) return;
    const int cv = (int)(idx % CV);
    long t = idx / CV;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const long b = t / Ho;
    float a[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) a[e] = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = oy + 1 - ky;
        if (ty < 0 || (ty & 1)) continue;
        const int i = ty >> 1;
        if (i >= H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = ox + 1 - kx;
            if (tx < 0 || (tx & 1)) continue;
            const int j = tx >> 1;
            if (j >= W) continue;
            const Vec16<CT> v = load16<CT>(cols + ((b * H + i) * W + j) * (9L * C) + (ky * 3 + kx) * C + cv * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] += v.get(e);
        }
    }
    T* o = out + ((b * Ho + oy) * Wo + ox) * C + cv * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) store_T(o + e, a[e]);
}

// ------------------------------------------------------------------------------------- xyz out layer
template <typename T>
__global__ __launch_bounds__(256) void xyz_out_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ out_nchw,
                                                      float* __restrict__ out_nhwc4, long rows, int HW, int C) {
    constexpr int VEC = Vec16<T>::N;
    const int chunks = C / VEC;                  // 16-B chunks per pixel
    const int LPP = chunks < 64 ? chunks : 64;   // lanes per pixel (power of two)
    const int lane = threadIdx.x & 63;
    const int sub = lane / LPP, l = lane % LPP;
    const int ppw = 64 / LPP;                    // pixels per wave per step
    const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long row = wave_id * ppw + sub;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    if (row < rows) {
        for (int c = l; c < chunks; c += LPP) {
            const Vec16<T> v = load16<T>(x + row * C + c * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float f = v.get(e);
                const int ch = c * VEC + e;
                a0 += f * w[ch];
                a1 += f * w[C + ch];
                a2 += f * w[2 * C + ch];
            }
        }
    }
    a0 = group_sum(a0, LPP); a1 = group_sum(a1, LPP); a2 = group_sum(a2, LPP);
    if (row < rows && l == 0) {
        a0 += bias[0]; a1 += bias[1]; a2 += bias[2];
        const long b = row / HW, pix = row - b * HW;
        out_nchw[(b * 3 + 0) * HW + pix] = a0;
        out_nchw[(b * 3 + 1) * HW + pix] = a1;
        out_nchw[(b * 3 + 2) * HW + pix] = a2;
        *reinterpret_cast<f32x4*>(out_nhwc4 + row * 4) = f32x4{a0, a1, a2, 0.f};
    }
}

// thread = one 16-B vector of output channels (its 3 x VEC weights + bias live in registers) x PXT pixels
template <typename T>
__global__ __launch_bounds__(256) void pointwise_k3_kernel(const float* __restrict__ xyz4, const float* __restrict__ w,
                                                           const float* __restrict__ bias, T* __restrict__ y,
                                                           long rows, int Cout) {
    constexpr int VEC = Vec16<T>::N, PXT = 8;
    const int CT = Cout / VEC, PG = 256 / CT;
    const int cs = threadIdx.x % CT, pl = threadIdx.x / CT;
    float w0[VEC], w1[VEC], w2[VEC], bv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int n = cs * VEC + e;
        w0[e] = w[n * 3]; w1[e] = w[n * 3 + 1]; w2[e] = w[n * 3 + 2]; bv[e] = bias[n];
    }
    const long r0 = (long)blockIdx.x * PG * PXT + pl;
#pragma unroll
    for (int i = 0; i < PXT; ++i) {
        const long row = r0 + (long)i * PG;
        if (row >= rows) break;
        const f32x4 p = *reinterpret_cast<const f32x4*>(xyz4 + row * 4);
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.set(e, fmaf(w0[e], p[0], fmaf(w1[e], p[1], fmaf(w2[e], p[2], bv[e]))));
        store16<T>(y + row * Cout + cs * VEC, o);
    }
}

// ------------------------------------------------------------------------------------- tiny-Cin 3x3 s2 conv
// input channels: c < 3 from xyz4 (B*R*R, 4); c in {3,4} from coord2d (B,2,R,R) when CIN == 5.
// wt: (CIN*9, Cout) fp32, k = ci*9 + kh*3 + kw.  Filter bank staged once in LDS; a thread owns 4 output channels of
// 4 consecutive output pixels of a row, so every 16-B filter read from LDS feeds 16 FMAs.
template <typename T, int CIN>
__global__ __launch_bounds__(256) void smallcin_conv3x3s2_kernel(const float* __restrict__ xyz4,
                                                                const float* __restrict__ coord2d,
                                                                const float* __restrict__ wt, T* __restrict__ y,
                                                                int B, int R, int Cout) {
    constexpr int KK = CIN * 9, PPT = 4;
    extern __shared__ __attribute__((aligned(16))) float w_s[];  // [KK][Cout]
    for (int i = threadIdx.x; i < KK * Cout; i += 256) w_s[i] = wt[i];
    __syncthreads();
    const int Ro = R / 2, CQ = Cout / 4, PB = 256 / CQ;          // PB pixel-quads per block
    const int cq = threadIdx.x % CQ;
    const long nquad = (long)B * Ro * (Ro / PPT);
    const long quad = (long)blockIdx.x * PB + threadIdx.x / CQ;
    if (quad >= nquad) return;
    const int wq = (int)(quad % (Ro / PPT));
    const long t = quad / (Ro / PPT);
    const int ho = (int)(t % Ro);
    const long b = t / Ro;
    const int wo0 = wq * PPT;
    float acc[PPT][4];
#pragma unroll
    for (int p = 0; p < PPT; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[p][e] = 0.f;
    for (int kh = 0; kh < 3; ++kh) {
        const int hi = ho * 2 - 1 + kh;
        if ((unsigned)hi >= (unsigned)R) continue;
        float in[2 * PPT + 1][CIN];                                // input columns wo0*2-1 .. wo0*2+2*PPT-1
#pragma unroll
        for (int c = 0; c < 2 * PPT + 1; ++c) {
            const int wi = wo0 * 2 - 1 + c;
            const bool ok = (unsigned)wi < (unsigned)R;
            const f32x4 p4 = ok ? *reinterpret_cast<const f32x4*>(xyz4 + ((b * R + hi) * R + wi) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
            in[c][0] = p4[0]; in[c][1] = p4[1]; in[c][2] = p4[2];
            if constexpr (CIN == 5) {
                in[c][3] = ok ? coord2d[((b * 2 + 0) * R + hi) * R + wi] : 0.f;
                in[c][4] = ok ? coord2d[((b * 2 + 1) * R + hi) * R + wi] : 0.f;
            }
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w_s + (c * 9 + kh * 3 + kw) * Cout + cq * 4);
#pragma unroll
                for (int p = 0; p < PPT; ++p)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[p][e] = fmaf(in[2 * p + kw][c], wv[e], acc[p][e]);
            }
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        T* o = y + ((b * Ro + ho) * Ro + wo0 + p) * Cout + cq * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) store_T(o + e, acc[p][e]);
    }
}

// The same convolution on the matrix pipe (fp16 output, Cout % 32 == 0, Ro % 32 == 0): two output rows (64 pixels) per
// workgroup, wave w owns channels [Cout/4 * w, ...) in 16-row A tiles.  K order inside the kernel: k' = tap * 8 + c (c padded
// to 8), so a lane's 8 k are the <= 5 channels of ONE tap = two 16-byte reads of the staged input rows [5 rows][R + 2][8];
// 96 = three 32-deep MFMA steps (taps 9-11 are zero).  fp32 inputs and taps are split into fp16 hi + lo and
// x_hi w_hi + x_lo w_hi + x_hi w_lo accumulates in fp32 (see stem_mfma_kernel): fp32-accurate like the VALU form.
template <int CIN, int NTW>   // NTW: 16-channel tiles per wave (Cout = 64 NTW)
__global__ __launch_bounds__(256) void smallcin_conv3x3s2_mfma_kernel(const float* __restrict__ xyz4,
                                                                     const float* __restrict__ coord2d,
                                                                     const float* __restrict__ wt, half_t* __restrict__ y,
                                                                     int B, int R) {
    constexpr int RMAX = 64;
    __shared__ __attribute__((aligned(16))) float in_s[5][RMAX + 2][8];
    const int Ro = R / 2, Cout = 64 * NTW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.x / (Ro / 2), ho0 = (blockIdx.x % (Ro / 2)) * 2;
    // ---- stage input rows 2 ho0 - 1 .. 2 ho0 + 3, columns -1 .. R (zero outside the image), 8 channels per pixel
    for (int i = tid; i < 5 * (R + 2); i += 256) {
        const int sr = i / (R + 2), sc = i - sr * (R + 2);
        const int hi = 2 * ho0 - 1 + sr, wi = sc - 1;
        const bool ok = (unsigned)hi < (unsigned)R && (unsigned)wi < (unsigned)R;
        f32x4 a = {0.f, 0.f, 0.f, 0.f}, c4 = {0.f, 0.f, 0.f, 0.f};
        if (ok) {
            a = *reinterpret_cast<const f32x4*>(xyz4 + (((long)b * R + hi) * R + wi) * 4);
            a[3] = 0.f;
            if constexpr (CIN == 5) {
                a[3] = coord2d[(((long)b * 2 + 0) * R + hi) * R + wi];
                c4[0] = coord2d[(((long)b * 2 + 1) * R + hi) * R + wi];
            }
        }
        *reinterpret_cast<f32x4*>(&in_s[sr][sc][0]) = a;
        *reinterpret_cast<f32x4*>(&in_s[sr][sc][4]) = c4;
    }
    // ---- A fragments: lane (fr, fq) of (nt, st): channel n = (wave * NTW + nt) * 16 + fr, k' = st*32 + fq*8 + j = tap (4 st + fq), c = j
    half8 ah[NTW][3], al[NTW][3];
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int tap = st * 4 + fq;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float w = (tap < 9 && j < CIN) ? wt[(j * 9 + tap) * Cout + (wave * NTW + nt) * 16 + fr] : 0.f;
                const half_t hi = (half_t)w;
                ah[nt][st][j] = hi;
                al[nt][st][j] = (half_t)(w - (float)hi);
            }
        }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int p = mt * 16 + fr, ro = p >> 5, wo = p & 31;   // pixel of this lane's B column (Ro == 32: one row = 2 tiles)
        half8 bh[3], bl[3];
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            const int tap = min(st * 4 + fq, 8), kh = tap / 3, kw = tap - kh * 3;
            const float* src = &in_s[2 * ro + kh][2 * wo + kw][0];
            const f32x4 x0 = *reinterpret_cast<const f32x4*>(src), x1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float x = j < 4 ? x0[j & 3] : x1[j & 3];
                const half_t hi = (half_t)x;
                bh[st][j] = hi;
                bl[st][j] = (half_t)(x - (float)hi);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[nt][st], bh[st], a, 0, 0, 0);   // small terms first
                a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nt][st], bl[st], a, 0, 0, 0);
            }
#pragma unroll
            for (int st = 0; st < 3; ++st) a = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[nt][st], bh[st], a, 0, 0, 0);
            half4 o = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3]};
            *reinterpret_cast<half4*>(y + (((long)b * Ro + ho0 + ro) * Ro + wo) * Cout + (wave * NTW + nt) * 16 + fq * 4) = o;
        }
    }
}

// ------------------------------------------------------------------------------------- ViT-style map encoder pieces
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ xyz4, T* __restrict__ out, int B, int R,
                                                       int P) {
    const int K = P * P * 3, np = R / P;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * np * np * P * P) return;
    const int pp = (int)(idx % (P * P));
    long t = idx / (P * P);
    const int px = (int)(t % np); t /= np;
    const int py = (int)(t % np);
    const long b = t / np;
    const int ky = pp / P, kx = pp - ky * P;
    const f32x4 v = *reinterpret_cast<const f32x4*>(xyz4 + ((b * R + py * P + ky) * R + px * P + kx) * 4);
    T* o = out + ((b * np + py) * np + px) * K + pp * 3;
    store_T(o, v[0]); store_T(o + 1, v[1]); store_T(o + 2, v[2]);
}

// one wavefront per (batch, head); lane = query token.  K and V of the head are staged in LDS as fp32; each lane
// keeps its 64 scores in registers (two-pass softmax), so nothing but q/k/v/out touches memory.
template <typename T>
__global__ __launch_bounds__(64) void attention64_kernel(const T* __restrict__ qkv, T* __restrict__ out, int heads) {
    __shared__ float ks[64][33], vs[64][33];
    const int b = blockIdx.x / heads, h = blockIdx.x - b * heads, i = threadIdx.x;
    const int C3 = 3 * heads * 32, C = heads * 32;
    const T* row = qkv + ((long)b * 64 + i) * C3 + h * 32;
    float q[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) {
        q[d] = (float)row[d] * 0.17677669529663687f;   // head_dim ** -0.5
        ks[i][d] = (float)row[C + d];
        vs[i][d] = (float)row[2 * C + d];
    }
    __syncthreads();
    float s[64], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        float a = 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) a = fmaf(q[d], ks[j][d], a);
        s[j] = a;
        mx = fmaxf(mx, a);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        s[j] = expf(s[j] - mx);
        sum += s[j];
    }
    const float inv = 1.0f / sum;
    float o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < 64; ++j) {
        const float pj = s[j] * inv;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = fmaf(pj, vs[j][d], o[d]);
    }
    T* orow = out + ((long)b * 64 + i) * C + h * 32;
#pragma unroll
    for (int d = 0; d < 32; ++d) store_T(orow + d, o[d]);
}

// ------------------------------------------------------------------------------------- ResNet stem / maxpool
// conv7x7 s2 p3 (3 -> 64, BN folded) + ReLU.  Block = 64 output pixels of one output row: the 7 x 133 x 3 input
// patch and the 147 x 64 filter bank are staged in LDS; thread = 4 output channels x 4 pixels.
template <typename T>
__global__ __launch_bounds__(256) void resnet_stem_kernel(const float* __restrict__ img, const float* __restrict__ wt,
                                                          const float* __restrict__ bias, T* __restrict__ out, int H,
                                                          int W) {
    constexpr int PXB = 64, IWD = 2 * PXB + 5;
    __shared__ __attribute__((aligned(16))) float w_s[147 * 64];
    __shared__ float in_s[3][7][IWD + 1];
    const int Ho = H / 2, Wo = W / 2, nwb = Wo / PXB;
    const int wblk = blockIdx.x % nwb, ho = (blockIdx.x / nwb) % Ho, b = blockIdx.x / (nwb * Ho);
    const int wo0 = wblk * PXB, tid = threadIdx.x;
    for (int i = tid; i < 147 * 64; i += 256) w_s[i] = wt[i];
    for (int i = tid; i < 3 * 7 * IWD; i += 256) {
        const int c = i / (7 * IWD), r = (i / IWD) % 7, x = i % IWD;
        const int hi = ho * 2 - 3 + r, wi = wo0 * 2 - 3 + x;
        in_s[c][r][x] = ((unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W) ? img[(((long)b * 3 + c) * H + hi) * W + wi] : 0.f;
    }
    __syncthreads();
    const int cq = tid & 15, pt = tid >> 4;
    float acc[4][4];
    const f32x4 bv = *reinterpret_cast<const f32x4*>(bias + cq * 4);
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[p][e] = bv[e];
    for (int c = 0; c < 3; ++c)
        for (int kh = 0; kh < 7; ++kh)
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w_s + ((c * 7 + kh) * 7 + kw) * 64 + cq * 4);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float v = in_s[c][kh][(pt * 4 + p) * 2 + kw];
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[p][e] = fmaf(v, wv[e], acc[p][e]);
                }
            }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        T* o = out + (((long)b * Ho + ho) * Wo + wo0 + pt * 4 + p) * 64 + cq * 4;
#pragma unroll
        for (int e = 0; e < 4; ++e) store_T(o + e, fmaxf(acc[p][e], 0.f));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H,
                                                           int W, int C) {
    constexpr int VEC = Vec16<T>::N;
    const int CT = C / VEC, Ho = H / 2, Wo = W / 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * Ho * Wo * CT) return;
    const int cs = (int)(idx % CT);
    long t = idx / CT;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const long b = t / Ho;
    float m[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) m[e] = -INFINITY;
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        if ((unsigned)iy >= (unsigned)H) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const Vec16<T> v = load16<T>(x + ((b * H + iy) * W + ix) * C + cs * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) m[e] = fmaxf(m[e], v.get(e));
        }
    }
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, m[e]);
    store16<T>(y + ((b * Ho + oy) * Wo + ox) * C + cs * VEC, o);
}

// ------------------------------------------------------------------------------------- SizeHead
// kernel 0: grid (B, C/256): AdaptiveMaxPool over HW -> pooled (B, C) fp32.  (Round 1 repeated this pool in every one of the
// F/16 workgroups of an image: 32 x the 128-KB map per image at bs 64, 40 us; now 8 MB once.)
template <typename T>
__global__ __launch_bounds__(256) void size_pool_kernel(const T* __restrict__ feat, float* __restrict__ pooled, int HW, int C) {
    const int b = blockIdx.x, cq = blockIdx.y * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;   // 4 channels per lane, 4 pixel quarters
    __shared__ float m_s[4][64][4];
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (cq * 4 < C) {
        const T* p = feat + (long)b * HW * C + cq * 4;
        for (int px = part; px < HW; px += 4) {
            if constexpr (sizeof(T) == 2) {
                const half4 h = *reinterpret_cast<const half4*>(p + (long)px * C);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], (float)h[e]);
            } else {
                const f32x4 h = *reinterpret_cast<const f32x4*>(p + (long)px * C);
#pragma unroll
                for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], h[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) m_s[part][threadIdx.x & 63][e] = m[e];
    __syncthreads();
    if (part == 0 && cq * 4 < C) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            pooled[(long)b * C + cq * 4 + e] = fmaxf(fmaxf(m_s[0][threadIdx.x][e], m_s[1][threadIdx.x][e]), fmaxf(m_s[2][threadIdx.x][e], m_s[3][threadIdx.x][e]));
    }
}

// kernel 1: grid (B, F/16): pooled row -> LDS, then 16 hidden units (BN folded, ReLU) -> hid (B,F)
__global__ __launch_bounds__(256) void size_hidden_kernel(const float* __restrict__ pooled_g, const float* __restrict__ w1,
                                                          const float* __restrict__ b1, float* __restrict__ hid, int C, int F) {
    extern __shared__ float pooled[];  // [C]
    const int b = blockIdx.x, tid = threadIdx.x;
    for (int c = tid; c < C; c += 256) pooled[c] = pooled_g[(long)b * C + c];
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const int f0 = blockIdx.y * 16 + wave * 4;
    for (int c = lane; c < C; c += 64) {          // four hidden units side by side: four independent load streams
        const float x = pooled[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = fmaf(x, w1[(long)(f0 + j) * C + c], a[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float v = group_sum(a[j], 64);
        if (lane == 0) hid[(long)b * F + f0 + j] = fmaxf(v + b1[f0 + j], 0.f);
    }
}

__global__ __launch_bounds__(64) void size_out_kernel(const float* __restrict__ hid, const float* __restrict__ w2,
                                                      const float* __restrict__ b2, const float* __restrict__ mean_size,
                                                      float* __restrict__ size, int F) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float ms[3] = {mean_size[b * 3], mean_size[b * 3 + 1], mean_size[b * 3 + 2]};
    const float nrm = sqrtf(ms[0] * ms[0] + ms[1] * ms[1] + ms[2] * ms[2]);
    for (int o = 0; o < 3; ++o) {
        float a = 0.f;
        for (int f = lane; f < F; f += 64) a += hid[(long)b * F + f] * w2[o * F + f];
        a = group_sum(a, 64);
        if (lane == 0) size[b * 3 + o] = a + b2[o] + ms[o] / nrm;
    }
}

// ------------------------------------------------------------------------------------- pose tail
__global__ __launch_bounds__(64) void pose_tail_kernel(const float* __restrict__ h, const float* __restrict__ hz,
                                                       int ldh, const float* __restrict__ w_r,
                                                       const float* __restrict__ b_r, const float* __restrict__ w_t,
                                                       const float* __restrict__ b_t, const float* __restrict__ w_z,
                                                       const float* __restrict__ b_z, const float* __restrict__ cam_K,
                                                       const float* __restrict__ bbox_center,
                                                       const float* __restrict__ resize_ratio,
                                                       const float* __restrict__ roi_wh, int wild6d, int site,
                                                       float* __restrict__ rot6d, float* __restrict__ pred_t,
                                                       float* __restrict__ rot_allo, float* __restrict__ rot_ego,
                                                       float* __restrict__ trans) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const f32x4 hv = *reinterpret_cast<const f32x4*>(h + (long)b * ldh + lane * 4);
    const f32x4 zv = *reinterpret_cast<const f32x4*>(hz + (long)b * ldh + lane * 4);
    float o[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const float* wrow = i < 6 ? w_r + i * 256 : (i < 8 ? w_t + (i - 6) * 256 : w_z);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + lane * 4);
        const f32x4 xv = i < 8 ? hv : zv;
        float a = wv[0] * xv[0] + wv[1] * xv[1] + wv[2] * xv[2] + wv[3] * xv[3];
        a = group_sum(a, 64);
        o[i] = a + (i < 6 ? b_r[i] : (i < 8 ? b_t[i - 6] : b_z[0]));
    }
    if (lane != 0) return;
    for (int i = 0; i < 6; ++i) rot6d[b * 6 + i] = o[i];
    for (int i = 0; i < 3; ++i) pred_t[b * 3 + i] = o[6 + i];
    // rot6d -> R (pose_utils/rot_reps.py:34-55), F.normalize eps 1e-12
    float x[3] = {o[0], o[1], o[2]}, yr[3] = {o[3], o[4], o[5]}, z[3], y[3];
    float n = fmaxf(sqrtf(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]), 1e-12f);
    for (int i = 0; i < 3; ++i) x[i] /= n;
    z[0] = x[1] * yr[2] - x[2] * yr[1]; z[1] = x[2] * yr[0] - x[0] * yr[2]; z[2] = x[0] * yr[1] - x[1] * yr[0];
    n = fmaxf(sqrtf(z[0] * z[0] + z[1] * z[1] + z[2] * z[2]), 1e-12f);
    for (int i = 0; i < 3; ++i) z[i] /= n;
    y[0] = z[1] * x[2] - z[2] * x[1]; y[1] = z[2] * x[0] - z[0] * x[2]; y[2] = z[0] * x[1] - z[1] * x[0];
    float Ra[9];
    for (int i = 0; i < 3; ++i) { Ra[i * 3] = x[i]; Ra[i * 3 + 1] = y[i]; Ra[i * 3 + 2] = z[i]; }
    for (int i = 0; i < 9; ++i) rot_allo[b * 9 + i] = Ra[i];
    // centroid / z back-projection (pose_from_pred_centroid_z.py:75-121)
    const float* K = cam_K + b * 9;
    const float c0 = site ? o[6] : o[6] * 0.f, c1 = site ? o[7] : o[7] * 0.f;
    const float cx = c0 * roi_wh[b * 2] + bbox_center[b * 2], cy = c1 * roi_wh[b * 2 + 1] + bbox_center[b * 2 + 1];
    float zz = o[8] * resize_ratio[b];
    if (wild6d) zz = zz * cam_K[0] / 590.f;
    const float tr[3] = {zz * (cx - K[2]) / K[0], zz * (cy - K[5]) / K[4], zz};
    for (int i = 0; i < 3; ++i) trans[b * 3 + i] = tr[i];
    // allocentric -> egocentric (pose_utils/utils.py:29-84): float32 ray, float64 rotation
    const float tn = sqrtf(tr[0] * tr[0] + tr[1] * tr[1] + tr[2] * tr[2]);
    const float ray[3] = {tr[0] / tn, tr[1] / tn, tr[2] / tn};
    const double angle = acos((double)ray[2]);
    if (angle > 0.0) {
        double ax = -(double)ray[1], ay = (double)ray[0], az = 0.0;  // cross((0,0,1), ray)
        const double an = sqrt(ax * ax + ay * ay + az * az);
        ax /= an; ay /= an; az /= an;
        const double c = cos(angle), s = sin(angle), Cc = 1.0 - c;
        const double xs = ax * s, ys = ay * s, zs = az * s, xC = ax * Cc, yC = ay * Cc, zC = az * Cc;
        const double xyC = ax * yC, yzC = ay * zC, zxC = az * xC;
        const double M[9] = {ax * xC + c, xyC - zs, zxC + ys, xyC + zs, ay * yC + c, yzC - xs,
                             zxC - ys, yzC + xs, az * zC + c};
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double a = 0.0;
                for (int k = 0; k < 3; ++k) a += M[i * 3 + k] * (double)Ra[k * 3 + j];
                rot_ego[b * 9 + i * 3 + j] = (float)a;
            }
    } else {
        for (int i = 0; i < 9; ++i) rot_ego[b * 9 + i] = Ra[i];
    }
}

__global__ void mask_resize_kernel(const float* __restrict__ m, float* __restrict__ out, int B, int S, int R) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)B * R * R) return;
    const int x = (int)(idx % R);
    const long t = idx / R;
    const int y = (int)(t % R);
    const long b = t / R;
    // legacy nearest: src = floor(dst * S / R)
    const int sy = min((int)floorf(y * ((float)S / R)), S - 1), sx = min((int)floorf(x * ((float)S / R)), S - 1);
    out[idx] = m[(b * S + sy) * S + sx];
}

}  // namespace

#define GP_DT_OK(dt) GP_REQUIRE((dt) == GP_F32 || (dt) == GP_F16, "bad dtype %d", (dt))

extern "C" int gp_convnext_stem(const float* img, const float* w, const float* b, const float* ln_w,
                                const float* ln_b, void* out, int B, int H, int W, int C0, float eps, int dtype,
                                void* stream) {
    GP_REQUIRE(img && w && b && ln_w && ln_b && out && B > 0, "gp_convnext_stem: bad argument");
    GP_DT_OK(dtype);
    GP_REQUIRE(C0 == 128, "gp_convnext_stem: C0=%d unsupported (128)", C0);
    GP_REQUIRE(H % 4 == 0 && W % 64 == 0, "gp_convnext_stem: H,W=%d,%d must be multiples of 4,64", H, W);
    const int PXB = (W / 4) % 64 == 0 ? 64 : 16;
    hipStream_t s = (hipStream_t)stream;
    const long px = (long)B * (H / 4) * (W / 4);
    gp_timing_before(s, GP_KC_SMALL, 2.0 * px * 48 * C0, (double)B * 3 * H * W * 4 + (double)px * C0 * (dtype == GP_F16 ? 2 : 4));
    dim3 grid(B * (H / 4) * (W / 4 / PXB));
    static const bool mfma_stem = [] { const char* e = getenv("GP_STEM_MFMA"); return !(e && e[0] == '0'); }();   // A/B switch
    // few crops (the detections of one frame): one output row per workgroup -- 64 workgroups per crop instead of 16 (GP_STEM_RPW=4 / 1 forces a form: A/B switch)
    static const int rpw_env = [] { const char* e = getenv("GP_STEM_RPW"); return e ? atoi(e) : 0; }();
    const bool rows4 = rpw_env ? rpw_env == 4 : (long)B * (H / 16) * (W / 4 / 64) >= 256;
    if (dtype == GP_F16 && PXB == 64 && mfma_stem && (H / 4) % 4 == 0 && rows4)
        hipLaunchKernelGGL(stem_mfma_kernel<4>, dim3(B * (H / 16) * (W / 4 / 64)), dim3(256), 0, s, img, w, b, ln_w, ln_b, (half_t*)out, H, W, eps);
    else if (dtype == GP_F16 && PXB == 64 && mfma_stem)
        hipLaunchKernelGGL(stem_mfma_kernel<1>, grid, dim3(256), 0, s, img, w, b, ln_w, ln_b, (half_t*)out, H, W, eps);
    else if (dtype == GP_F16) hipLaunchKernelGGL(stem_kernel<half_t>, grid, dim3(256), 0, s, img, w, b, ln_w, ln_b, (half_t*)out, H, W, eps, PXB);
    else hipLaunchKernelGGL(stem_kernel<float>, grid, dim3(256), 0, s, img, w, b, ln_w, ln_b, (float*)out, H, W, eps, PXB);
    GP_LAUNCH_CHECK("gp_convnext_stem");
}

extern "C" int gp_upsample_bilinear2x(const void* x, void* y, int B, int H, int W, int C, int dtype_in, void* stream) {
    GP_REQUIRE(x && y && B > 0 && H > 1 && W > 1, "gp_upsample_bilinear2x: bad argument");
    const int dtype = dtype_in & ~GP_OUT_PLANES;
    GP_DT_OK(dtype);
    const long pl = (dtype_in & GP_OUT_PLANES) ? (long)B * 4 * H * W * C : 0;
    const int esz = dtype == GP_F16 ? 2 : 4, vec = 16 / esz;
    GP_REQUIRE(C % vec == 0, "gp_upsample_bilinear2x: C=%d", C);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * 4 * H * W * (C / vec);
    gp_timing_before(s, GP_KC_ELEMENTWISE, 8.0 * total * vec, (double)B * H * W * C * esz * 5);
    const int ct = C / vec;
    GP_REQUIRE(!pl || (dtype == GP_F32 && (ct & (ct - 1)) == 0 && 2 * H <= 65535 && B <= 65535), "gp_upsample_bilinear2x: GP_OUT_PLANES needs GP_F32 and C / 4 a power of two");
    if ((ct & (ct - 1)) == 0 && 2 * H <= 65535 && B <= 65535) {
        int sh = 0;
        while ((1 << sh) < ct) ++sh;
        dim3 grid(cdiv((long)2 * W * ct, 256), 2 * H, B);
        if (dtype == GP_F16) hipLaunchKernelGGL(upsample2x_rows_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)x, (half_t*)y, H, W, C, sh, 0l);
        else hipLaunchKernelGGL(upsample2x_rows_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)y, H, W, C, sh, pl);
        GP_LAUNCH_CHECK("gp_upsample_bilinear2x");
    }
    if (dtype == GP_F16) hipLaunchKernelGGL(upsample2x_kernel<half_t>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const half_t*)x, (half_t*)y, B, H, W, C);
    else hipLaunchKernelGGL(upsample2x_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, C);
    GP_LAUNCH_CHECK("gp_upsample_bilinear2x");
}

extern "C" int gp_crop_rois(const unsigned char* frames, const unsigned char* masks, const int* frame_idx, const int* mask_idx,
                            const double* inv_img, const double* inv_out, const float* img_lut, const float* xlut,
                            const float* ylut, float* roi_img, float* roi_mask, float* roi_coord_2d, int B, int F, int NM,
                            int H, int W, int S, int R, void* stream) {
    GP_REQUIRE(frames && masks && frame_idx && mask_idx && inv_img && inv_out && img_lut && xlut && ylut && roi_img && roi_mask && roi_coord_2d,
               "gp_crop_rois: null pointer");
    GP_REQUIRE(B > 0 && B <= 65535 && F > 0 && NM > 0 && H > 0 && W > 0 && S > 0 && R > 0 && R <= S, "gp_crop_rois: bad shape");
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 0.0, (double)B * (S * S * 4.0 * 4 + S * S * 4.0 + R * R * 8.0));
    hipLaunchKernelGGL(crop_rois_kernel, dim3(cdiv(S * S, 256), B), dim3(256), 0, s, frames, masks, frame_idx, mask_idx, inv_img,
                       inv_out, img_lut, xlut, ylut, roi_img, roi_mask, roi_coord_2d, H, W, S, R);
    GP_LAUNCH_CHECK("gp_crop_rois");
}

extern "C" int gp_pred_rt(const float* R, const float* t, const float* size, const float* scale, float* pred_rt,
                          float* pred_size, int B, void* stream) {
    GP_REQUIRE(R && t && size && pred_rt && pred_size && B > 0, "gp_pred_rt: bad argument");
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 0.0, (double)B * 35 * 4);
    hipLaunchKernelGGL(pred_rt_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, R, t, size, scale, pred_rt, pred_size, B);
    GP_LAUNCH_CHECK("gp_pred_rt");
}

extern "C" int gp_deconv_col2im(const void* cols, void* out, int B, int H, int W, int C, int dtype_in, void* stream) {
    const int dtype = dtype_in & ~GP_COLS_F16;
    const bool c16 = (dtype_in & GP_COLS_F16) != 0;
    GP_REQUIRE(cols && out && B > 0 && C % (c16 ? 8 : 4) == 0, "gp_deconv_col2im: bad argument");
    GP_DT_OK(dtype);
    GP_REQUIRE(!c16 || dtype == GP_F16, "gp_deconv_col2im: fp16 cols go with fp16 output");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * 4 * H * W * (C / (c16 ? 8 : 4));
    gp_timing_before(s, GP_KC_ELEMENTWISE, (double)B * H * W * 9 * C, (double)B * H * W * 9 * C * (c16 ? 2 : 4) + (double)B * 4 * H * W * C * (dtype == GP_F16 ? 2 : 4));
    if (c16) hipLaunchKernelGGL((col2im_kernel<half_t, half_t>), dim3(cdiv(total, 256)), dim3(256), 0, s, (const half_t*)cols, (half_t*)out, B, H, W, C);
    else if (dtype == GP_F16) hipLaunchKernelGGL((col2im_kernel<half_t, float>), dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)cols, (half_t*)out, B, H, W, C);
    else hipLaunchKernelGGL((col2im_kernel<float, float>), dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)cols, (float*)out, B, H, W, C);
    GP_LAUNCH_CHECK("gp_deconv_col2im");
}

extern "C" int gp_xyz_out_layer(const void* x, const float* w, const float* b, float* out_nchw, float* out_nhwc4,
                                int B, int HW, int C, int dtype, void* stream) {
    GP_REQUIRE(x && w && b && out_nchw && out_nhwc4 && B > 0, "gp_xyz_out_layer: bad argument");
    GP_DT_OK(dtype);
    const int esz = dtype == GP_F16 ? 2 : 4, vec = 16 / esz, chunks = C / vec;
    GP_REQUIRE(C % vec == 0 && (chunks & (chunks - 1)) == 0, "gp_xyz_out_layer: C=%d", C);
    const int LPP = chunks < 64 ? chunks : 64, ppw = 64 / LPP;
    const long rows = (long)B * HW;
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 6.0 * rows * C, (double)rows * C * esz + rows * 28.0);
    dim3 grid(cdiv(rows, 4L * ppw));
    if (dtype == GP_F16) hipLaunchKernelGGL(xyz_out_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)x, w, b, out_nchw, out_nhwc4, rows, HW, C);
    else hipLaunchKernelGGL(xyz_out_kernel<float>, grid, dim3(256), 0, s, (const float*)x, w, b, out_nchw, out_nhwc4, rows, HW, C);
    GP_LAUNCH_CHECK("gp_xyz_out_layer");
}

extern "C" int gp_pointwise_k3(const float* xyz4, const float* w, const float* b, void* y, long rows, int Cout,
                               int dtype, void* stream) {
    GP_REQUIRE(xyz4 && w && b && y && rows > 0, "gp_pointwise_k3: bad argument");
    GP_DT_OK(dtype);
    const int esz = dtype == GP_F16 ? 2 : 4, vec = 16 / esz;
    GP_REQUIRE(Cout % vec == 0 && 256 % (Cout / vec) == 0, "gp_pointwise_k3: Cout=%d", Cout);
    hipStream_t s = (hipStream_t)stream;
    const long total = cdiv(rows, (256 / (Cout / vec)) * 8) * 256;
    gp_timing_before(s, GP_KC_ELEMENTWISE, 6.0 * rows * Cout, rows * 16.0 + (double)rows * Cout * esz);
    if (dtype == GP_F16) hipLaunchKernelGGL(pointwise_k3_kernel<half_t>, dim3(cdiv(total, 256)), dim3(256), 0, s, xyz4, w, b, (half_t*)y, rows, Cout);
    else hipLaunchKernelGGL(pointwise_k3_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, s, xyz4, w, b, (float*)y, rows, Cout);
    GP_LAUNCH_CHECK("gp_pointwise_k3");
}

template <int CIN>
static int launch_smallcin(const float* xyz4, const float* coord2d, const float* w, void* y, int B, int R, int Cout,
                           int dtype, void* stream, const char* name) {
    GP_REQUIRE(xyz4 && w && y && B > 0 && R % 8 == 0, "%s: bad argument", name);
    GP_DT_OK(dtype);
    GP_REQUIRE(Cout % 4 == 0 && Cout / 4 <= 256 && 256 % (Cout / 4) == 0, "%s: Cout=%d", name, Cout);
    hipStream_t s = (hipStream_t)stream;
    const long pix = (long)B * (R / 2) * (R / 2);
        const size_t lds = (size_t)CIN * 9 * Cout * sizeof(float);
    gp_timing_before(s, GP_KC_SMALL, 2.0 * pix * CIN * 9 * Cout, (double)B * R * R * CIN * 4 + (double)pix * Cout * (dtype == GP_F16 ? 2 : 4));
    static const bool mfma = [] { const char* e = getenv("GP_SMALLCIN_MFMA"); return !(e && e[0] == '0'); }();   // A/B switch
    if (dtype == GP_F16 && mfma && R == 64 && (Cout == 128 || Cout == 256)) {   // 64 x 64 maps -> 32 x 32: two output rows per workgroup
        if (Cout == 128) hipLaunchKernelGGL((smallcin_conv3x3s2_mfma_kernel<CIN, 2>), dim3(B * (R / 4)), dim3(256), 0, s, xyz4, coord2d, w, (half_t*)y, B, R);
        else hipLaunchKernelGGL((smallcin_conv3x3s2_mfma_kernel<CIN, 4>), dim3(B * (R / 4)), dim3(256), 0, s, xyz4, coord2d, w, (half_t*)y, B, R);
    } else if (dtype == GP_F16)
        hipLaunchKernelGGL((smallcin_conv3x3s2_kernel<half_t, CIN>), dim3(cdiv(pix / 4, 256 / (Cout / 4))), dim3(256), lds, s, xyz4, coord2d, w, (half_t*)y, B, R, Cout);
    else
        hipLaunchKernelGGL((smallcin_conv3x3s2_kernel<float, CIN>), dim3(cdiv(pix / 4, 256 / (Cout / 4))), dim3(256), lds, s, xyz4, coord2d, w, (float*)y, B, R, Cout);
    GP_LAUNCH_CHECK(name);
}

extern "C" int gp_pnp_conv1(const float* xyz4, const float* coord2d, const float* w, void* y, int B, int R,
                            int Cout, int dtype, void* stream) {
    GP_REQUIRE(coord2d != nullptr, "gp_pnp_conv1: null coord2d");
    return launch_smallcin<5>(xyz4, coord2d, w, y, B, R, Cout, dtype, stream, "gp_pnp_conv1");
}

extern "C" int gp_xyz_conv3x3_s2(const float* xyz4, const float* w, void* y, int B, int R, int Cout, int dtype,
                                 void* stream) {
    return launch_smallcin<3>(xyz4, nullptr, w, y, B, R, Cout, dtype, stream, "gp_xyz_conv3x3_s2");
}

extern "C" int gp_size_head(const void* feat, const float* w1, const float* b1, const float* w2, const float* b2,
                            const float* mean_size, float* size, float* scratch, int B, int HW, int C, int F,
                            int dtype, void* stream) {
    GP_REQUIRE(feat && w1 && b1 && w2 && b2 && mean_size && size && scratch && B > 0, "gp_size_head: bad argument");
    GP_DT_OK(dtype);
    GP_REQUIRE(C % 4 == 0 && F % 16 == 0, "gp_size_head: C=%d F=%d", C, F);
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 2.0 * B * (C * F + 3 * F), (double)B * HW * C * (dtype == GP_F16 ? 2 : 4));
    const size_t lds = (size_t)C * sizeof(float);
    float* pooled = scratch + (long)B * F;         // scratch: (B, F) hidden units, then (B, C) pooled maxima
    if (dtype == GP_F16) hipLaunchKernelGGL(size_pool_kernel<half_t>, dim3(B, cdiv(C, 256)), dim3(256), 0, s, (const half_t*)feat, pooled, HW, C);
    else hipLaunchKernelGGL(size_pool_kernel<float>, dim3(B, cdiv(C, 256)), dim3(256), 0, s, (const float*)feat, pooled, HW, C);
    hipLaunchKernelGGL(size_hidden_kernel, dim3(B, F / 16), dim3(256), lds, s, pooled, w1, b1, scratch, C, F);
    hipLaunchKernelGGL(size_out_kernel, dim3(B), dim3(64), 0, s, scratch, w2, b2, mean_size, size, F);
    GP_LAUNCH_CHECK("gp_size_head");
}

extern "C" int gp_pose_tail(const float* h, const float* hz, int ldh, const float* w_r, const float* b_r,
                            const float* w_t, const float* b_t, const float* w_z, const float* b_z,
                            const float* cam_K, const float* bbox_center, const float* resize_ratio,
                            const float* roi_wh, int wild6d, int site_centroid, float* rot6d, float* pred_t,
                            float* rot_allo, float* rot_ego, float* trans, int B, void* stream) {
    GP_REQUIRE(h && hz && w_r && b_r && w_t && b_t && w_z && b_z && cam_K && bbox_center && resize_ratio && roi_wh &&
                   rot6d && pred_t && rot_allo && rot_ego && trans && B > 0 && ldh >= 256 && ldh % 4 == 0,
               "gp_pose_tail: bad argument");
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 2.0 * B * 9 * 256, B * 2048.0);
    hipLaunchKernelGGL(pose_tail_kernel, dim3(B), dim3(64), 0, s, h, hz, ldh, w_r, b_r, w_t, b_t, w_z, b_z, cam_K,
                       bbox_center, resize_ratio, roi_wh, wild6d, site_centroid, rot6d, pred_t, rot_allo, rot_ego, trans);
    GP_LAUNCH_CHECK("gp_pose_tail");
}

extern "C" int gp_patchify_xyz(const float* xyz4, void* out, int B, int R, int P, int dtype, void* stream) {
    GP_REQUIRE(xyz4 && out && B > 0 && P > 0 && R % P == 0, "gp_patchify_xyz: bad argument");
    GP_DT_OK(dtype);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * R * R;
    gp_timing_before(s, GP_KC_ELEMENTWISE, 0.0, total * (16.0 + 3 * (dtype == GP_F16 ? 2 : 4)));
    if (dtype == GP_F16) hipLaunchKernelGGL(patchify_kernel<half_t>, dim3(cdiv(total, 256)), dim3(256), 0, s, xyz4, (half_t*)out, B, R, P);
    else hipLaunchKernelGGL(patchify_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, s, xyz4, (float*)out, B, R, P);
    GP_LAUNCH_CHECK("gp_patchify_xyz");
}

extern "C" int gp_attention64(const void* qkv, void* out, int B, int heads, int dtype, void* stream) {
    GP_REQUIRE(qkv && out && B > 0 && heads > 0, "gp_attention64: bad argument");
    GP_DT_OK(dtype);
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 4.0 * B * heads * 64 * 64 * 32, (double)B * 64 * heads * 32 * 4 * (dtype == GP_F16 ? 2 : 4));
    if (dtype == GP_F16) hipLaunchKernelGGL(attention64_kernel<half_t>, dim3(B * heads), dim3(64), 0, s, (const half_t*)qkv, (half_t*)out, heads);
    else hipLaunchKernelGGL(attention64_kernel<float>, dim3(B * heads), dim3(64), 0, s, (const float*)qkv, (float*)out, heads);
    GP_LAUNCH_CHECK("gp_attention64");
}

extern "C" int gp_resnet_stem(const float* img, const float* w, const float* b, void* out, int B, int H, int W, int dtype,
                              void* stream) {
    GP_REQUIRE(img && w && b && out && B > 0, "gp_resnet_stem: bad argument");
    GP_DT_OK(dtype);
    GP_REQUIRE(H % 2 == 0 && W % 128 == 0, "gp_resnet_stem: H,W=%d,%d must be multiples of 2,128", H, W);
    hipStream_t s = (hipStream_t)stream;
    const long px = (long)B * (H / 2) * (W / 2);
    gp_timing_before(s, GP_KC_SMALL, 2.0 * px * 147 * 64, (double)B * 3 * H * W * 4 + (double)px * 64 * (dtype == GP_F16 ? 2 : 4));
    dim3 grid(B * (H / 2) * (W / 2 / 64));
    if (dtype == GP_F16) hipLaunchKernelGGL(resnet_stem_kernel<half_t>, grid, dim3(256), 0, s, img, w, b, (half_t*)out, H, W);
    else hipLaunchKernelGGL(resnet_stem_kernel<float>, grid, dim3(256), 0, s, img, w, b, (float*)out, H, W);
    GP_LAUNCH_CHECK("gp_resnet_stem");
}

extern "C" int gp_maxpool3x3s2(const void* x, void* y, int B, int H, int W, int C, int dtype, void* stream) {
    GP_REQUIRE(x && y && B > 0 && H % 2 == 0 && W % 2 == 0, "gp_maxpool3x3s2: bad argument");
    GP_DT_OK(dtype);
    const int esz = dtype == GP_F16 ? 2 : 4, vec = 16 / esz;
    GP_REQUIRE(C % vec == 0, "gp_maxpool3x3s2: C=%d", C);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * (H / 2) * (W / 2) * (C / vec);
    gp_timing_before(s, GP_KC_ELEMENTWISE, 0.0, (double)B * H * W * C * esz * 1.25);
    if (dtype == GP_F16) hipLaunchKernelGGL(maxpool3x3s2_kernel<half_t>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const half_t*)x, (half_t*)y, B, H, W, C);
    else hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const float*)x, (float*)y, B, H, W, C);
    GP_LAUNCH_CHECK("gp_maxpool3x3s2");
}

extern "C" int gp_mask_resize_nearest(const float* mask, float* out, int B, int S, int R, void* stream) {
    GP_REQUIRE(mask && out && B > 0 && S > 0 && R > 0, "gp_mask_resize_nearest: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * R * R;
    gp_timing_before(s, GP_KC_ELEMENTWISE, 0.0, total * 8.0);
    hipLaunchKernelGGL(mask_resize_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, mask, out, B, S, R);
    GP_LAUNCH_CHECK("gp_mask_resize_nearest");
}

// =====================================================================================================
// fp32 rows -> the hi / lo' fp16 planes of the split-operand GEMM mode (gp_gemm_desc.split_shift): hi = fp16(x),
// lo' = fp16((x - hi) * 2^S).  x - hi is exact in fp32 (hi has 11 of x's leading bits), the scale is a power of two, so
// the only roundings are the two conversions: x = hi + 2^-S lo' to 2^-22 relative down to |x| ~ 2^-(14+S).
// 8 elements per lane: two 16-byte loads, one 16-byte store per plane.
namespace {
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ x, half_t* __restrict__ hi, half_t* __restrict__ lo,
                                                           long rows, int cols8, long ldx, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * cols8) return;
    const long r = i / cols8;
    const int c = (int)(i - r * cols8) * 8;
    const float* src = x + r * ldx + c;
    const f32x4 a = *reinterpret_cast<const f32x4*>(src), b = *reinterpret_cast<const f32x4*>(src + 4);
    half8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = j < 4 ? a[j] : b[j - 4];
        const half_t hv = (half_t)v;
        h[j] = hv;
        l[j] = (half_t)((v - (float)hv) * scale);
    }
    *reinterpret_cast<half8*>(hi + i * 8) = h;
    *reinterpret_cast<half8*>(lo + i * 8) = l;
}
}  // namespace

extern "C" int gp_split_planes(const float* x, void* planes, long rows, int cols, long ldx, long plane_stride, int split_shift,
                               void* stream) {
    GP_REQUIRE(x && planes, "gp_split_planes: null pointer");
    GP_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && ldx >= cols && ldx % 4 == 0, "gp_split_planes: bad shape rows=%ld cols=%d ldx=%ld", rows, cols, ldx);
    GP_REQUIRE(plane_stride >= rows * cols && plane_stride % 8 == 0, "gp_split_planes: plane_stride too small / unaligned");
    GP_REQUIRE(split_shift > 0 && split_shift <= 14, "gp_split_planes: split_shift in 1..14");
    GP_REQUIRE(((size_t)x & 15) == 0 && ((size_t)planes & 15) == 0, "gp_split_planes: 16-byte alignment");
    hipStream_t s = (hipStream_t)stream;
    const long total = rows * (cols / 8);
    gp_timing_before(s, GP_KC_ELEMENTWISE, 0.0, (double)rows * cols * 8.0);
    gp_timing_label("split_planes %ldx%d", rows, cols);
    half_t* hi = reinterpret_cast<half_t*>(planes);
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, x, hi, hi + plane_stride, rows, cols / 8, ldx,
                       ldexpf(1.0f, split_shift));
    GP_LAUNCH_CHECK("gp_split_planes");
}


// =====================================================================================================
// (R 9, t 3, s 3) -> one (B, 15) fp32 row per crop: the payload of the all-gather (givepose_amd/dist.py).  A library kernel
// rather than three strided PyTorch copies, so that everything that runs on the communication stream beside the MFMA kernels
// of the other slots is built with this library's flags.
namespace {
__global__ __launch_bounds__(256) void pack_poses_kernel(const float* __restrict__ R, const float* __restrict__ t, const float* __restrict__ sz,
                                                         float* __restrict__ out, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * 15) return;
    const int b = i / 15, c = i - b * 15;
    out[i] = c < 9 ? R[b * 9 + c] : c < 12 ? t[b * 3 + c - 9] : sz[b * 3 + c - 12];
}
}  // namespace

extern "C" int gp_pack_poses(const float* R, const float* t, const float* size, float* out, int B, void* stream) {
    GP_REQUIRE(R && t && size && out && B > 0, "gp_pack_poses: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_SMALL, 0.0, B * 120.0);
    hipLaunchKernelGGL(pack_poses_kernel, dim3(cdiv(B * 15, 256)), dim3(256), 0, s, R, t, size, out, B);
    GP_LAUNCH_CHECK("gp_pack_poses");
}
