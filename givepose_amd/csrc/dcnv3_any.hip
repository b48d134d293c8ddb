// DCNv3 forward for ANY geometry / dtype, and the DCNv3 backward (gfx950).
//
// The hot path (csrc/dcnv3.hip) covers what PoseNet launches: square 3x3 kernels, D % 4 == 0, fp16 / fp32.  The
// reference operator is wider -- dcnv3_cuda.cu:68 dispatches double / float / half, every geometry parameter exists per
// axis, any group_channels, and there is a backward -- and its own test (network/ops_dcnv3/test.py:35-170, 262-265)
// exercises double and channel counts such as 1, 30, 71.  These two kernels are that breadth:
//   * dcnv3_any_fwd_kernel: the arithmetic of dcnv3_im2col_cuda.cuh:216-282 (+ :32-80) with opmath = double for double,
//     float otherwise.  One WAVEFRONT per (output pixel, group): lane t resolves tap t once, the tap loop broadcasts it and
//     the lanes walk the D channels (coalesced runs);
//   * dcnv3_any_bwd_kernel: dcnv3_im2col_cuda.cuh:386-487 (+ :82-140).  One WAVEFRONT per (output pixel, group): the 64
//     lanes walk the D channels; grad_input is scattered with atomics (as the reference does), the per-tap grad_offset /
//     grad_mask partials are reduced over the channels with wave shuffles (the reference: a shared-memory tree over a
//     block of D threads) and written once -- no atomics on them, every (pixel, group, tap) has exactly one writer.
//   Gradient buffers are opmath typed (float for half / float inputs, double for double), zero-filled here
//   (dcnv3_cuda.cu:128-130 allocates them with zeros_like), so the unconsumed tail of a stride-2 offset buffer stays 0.
#include "common.hpp"

namespace {

struct AnyKP {
    const void *in, *off, *mask, *gout;
    void *out, *gin, *goff, *gmask;
    int N, H, W, G, D, kh, kw, sh, sw, ph, pw, dh, dw, rc, Ho, Wo;
    double os;
    long rows;   // N * Ho * Wo
};

// Forward, any geometry / dtype.  One WAVEFRONT per (output pixel, group) -- the mapping of the backward below and of the hot
// kernel in dcnv3.hip, not the reference's thread per output scalar: the sampling geometry of a (pixel, group, tap) does not
// depend on the channel, so lane t works out tap t ONCE per wave -- location, the four corner weights (zero for a corner
// outside the image, the fetch then goes to a clamped address, so there is no branch between the loads), the mask weight and
// the element offsets of the corners -- and the tap loop broadcasts those eight numbers from the owner lane (the lane index
// is wave-uniform: v_readlane, no LDS) while the 64 lanes walk the group's D channels in coalesced runs.  Taps are
// accumulated in the reference's order (kernel_w outer, kernel_h inner; dcnv3_im2col_cuda.cuh:248-276) in opmath A.
// More than 64 taps (kernels above 8x8) go through the owner lanes in chunks of 64; D > 64 walks channel chunks.
template <typename T, typename A>
__global__ __launch_bounds__(256) void dcnv3_any_fwd_kernel(const AnyKP p) {
    const long wid = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wid >= p.rows * p.G) return;                                   // wave-uniform
    const int g = (int)(wid % p.G);
    const long pix = wid / p.G;
    const int wo = (int)(pix % p.Wo), ho = (int)((pix / p.Wo) % p.Ho), b = (int)(pix / ((long)p.Wo * p.Ho));
    const int P = p.kh * p.kw - p.rc, centre = (p.kw / 2) * p.kh + p.kh / 2;
    const A os = (A)p.os;
    // dcnv3_im2col_cuda.cuh:236-244: p0 = centre of the undeformed window, p0_ = p0 - half window * offset_scale
    const int hw_ = (p.dw * (p.kw - 1)) >> 1, hh_ = (p.dh * (p.kh - 1)) >> 1;
    const A p0_w_ = (A)(hw_ - p.pw + wo * p.sw) - (A)hw_ * os;
    const A p0_h_ = (A)(hh_ - p.ph + ho * p.sh) - (A)hh_ * os;
    const long cs = (long)p.G * p.D;
    const T* off = reinterpret_cast<const T*>(p.off) + wid * P * 2;      // flat buffers, consumed row by row (:226-234)
    const T* msk = reinterpret_cast<const T*>(p.mask) + wid * P;
    const T* im = reinterpret_cast<const T*>(p.in) + (long)b * p.H * p.W * cs + (long)g * p.D;
    T* out = reinterpret_cast<T*>(p.out) + wid * p.D;
    A w1 = 0, w2 = 0, w3 = 0, w4 = 0, mw = 0;     // lane t: tap (q0 + t)
    long o11 = 0, ex = 0, ey = 0;       // element offsets (long: W * G * D overflows an int on large tensors)
    auto own_tap = [&](int q0) {
        const int q = q0 + lane;
        w1 = w2 = w3 = w4 = mw = 0;
        o11 = 0; ex = ey = 0;
        if (q < P) {
            const int qq = (p.rc && q >= centre) ? q + 1 : q;          // window position of tap q when the centre is removed
            const int i = qq / p.kh, j = qq - i * p.kh;
            const A lw_ = p0_w_ + ((A)(i * p.dw) + (A)off[2 * q]) * os;
            const A lh_ = p0_h_ + ((A)(j * p.dh) + (A)off[2 * q + 1]) * os;
            if (lh_ > (A)-1 && lw_ > (A)-1 && lh_ < (A)p.H && lw_ < (A)p.W) {
                const A fh = floor(lh_), fw = floor(lw_);
                const int y0 = (int)fh, x0 = (int)fw, y1 = y0 + 1, x1 = x0 + 1;
                const A lh = lh_ - fh, lw = lw_ - fw, hh = (A)1 - lh, hw = (A)1 - lw;
                const bool t = y0 >= 0, bo = y1 <= p.H - 1, l = x0 >= 0, r = x1 <= p.W - 1;
                w1 = (t && l) ? hh * hw : (A)0;
                w2 = (t && r) ? hh * lw : (A)0;
                w3 = (bo && l) ? lh * hw : (A)0;
                w4 = (bo && r) ? lh * lw : (A)0;
                const int cy0 = max(y0, 0), cy1 = min(y1, p.H - 1), cx0 = max(x0, 0), cx1 = min(x1, p.W - 1);
                o11 = ((long)cy0 * p.W + cx0) * cs;
                ex = (long)(cx1 - cx0) * cs;
                ey = (long)(cy1 - cy0) * p.W * cs;
                mw = (A)msk[q];
            }
        }
    };
    for (int c0 = 0; c0 < p.D; c0 += 64) {
        const int c = c0 + lane;
        A col = 0;
        for (int q0 = 0; q0 < P; q0 += 64) {
            if (c0 == 0 || P > 64) own_tap(q0);
            const int n = min(64, P - q0);
            for (int s = 0; s < n; ++s) {
                const A a1 = __shfl(w1, s, 64), a2 = __shfl(w2, s, 64), a3 = __shfl(w3, s, 64), a4 = __shfl(w4, s, 64);
                const A wg = __shfl(mw, s, 64);
                const long o = __shfl(o11, s, 64);
                const long dx = __shfl(ex, s, 64), dy = __shfl(ey, s, 64);
                if (c < p.D) {
                    const T* q = im + o + c;
                    // a corner outside the image is fetched from a clamped address: SELECT it away (the reference skips the load,
                    // cuh:50-68) -- a multiplication by the zero weight would let a NaN / Inf at the clamped pixel through
                    const A v1 = a1 != (A)0 ? (A)q[0] : (A)0, v2 = a2 != (A)0 ? (A)q[dx] : (A)0;
                    const A v3 = a3 != (A)0 ? (A)q[dy] : (A)0, v4 = a4 != (A)0 ? (A)q[dx + dy] : (A)0;
                    col += (a1 * v1 + a2 * v2 + a3 * v3 + a4 * v4) * wg;
                }
            }
        }
        if (c < p.D) out[c] = (T)col;
    }
}

template <typename A> __device__ __forceinline__ A wave_sum(A v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, typename A>
__global__ __launch_bounds__(256) void dcnv3_any_bwd_kernel(const AnyKP p) {
    const long wid = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;       // one wavefront per (pixel, group)
    const int lane = threadIdx.x & 63;
    if (wid >= p.rows * p.G) return;                                   // wave-uniform
    long t = wid;
    const int g = (int)(t % p.G); t /= p.G;
    const int p0_w = ((p.dw * (p.kw - 1)) >> 1) - p.pw + (int)(t % p.Wo) * p.sw; t /= p.Wo;
    const int p0_h = ((p.dh * (p.kh - 1)) >> 1) - p.ph + (int)(t % p.Ho) * p.sh; t /= p.Ho;
    const int b = (int)t;
    const int P = p.kh * p.kw - p.rc;
    long wptr = wid * P, lptr = wptr << 1;
    const T* off = reinterpret_cast<const T*>(p.off);
    const T* msk = reinterpret_cast<const T*>(p.mask);
    const T* gout = reinterpret_cast<const T*>(p.gout) + wid * p.D;
    const int w_stride = p.G * p.D, h_stride = p.W * w_stride;
    const T* im = reinterpret_cast<const T*>(p.in) + (long)b * p.H * h_stride + g * p.D;
    A* gim = reinterpret_cast<A*>(p.gin) + (long)b * p.H * h_stride + g * p.D;
    A* goff = reinterpret_cast<A*>(p.goff);
    A* gmask = reinterpret_cast<A*>(p.gmask);
    const A os = (A)p.os;
    const A p0_w_ = (A)p0_w - (A)((p.dw * (p.kw - 1)) >> 1) * os;
    const A p0_h_ = (A)p0_h - (A)((p.dh * (p.kh - 1)) >> 1) * os;
    for (int i = 0; i < p.kw; ++i)
        for (int j = 0; j < p.kh; ++j) {
            if (p.rc && i == p.kw / 2 && j == p.kh / 2) continue;
            const A w = p0_w_ + ((A)(i * p.dw) + (A)off[lptr]) * os;
            const A h = p0_h_ + ((A)(j * p.dh) + (A)off[lptr + 1]) * os;
            const A m = (A)msk[wptr];
            A gm = 0, gw = 0, gh = 0;
            if (h > (A)-1 && w > (A)-1 && h < (A)p.H && w < (A)p.W) {        // wave-uniform
                const int h_low = (int)floor(h), w_low = (int)floor(w), h_high = h_low + 1, w_high = w_low + 1;
                const A lh = h - (A)h_low, lw = w - (A)w_low, hh = (A)1 - lh, hw = (A)1 - lw;
                const A w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                const bool o1 = h_low >= 0 && w_low >= 0, o2 = h_low >= 0 && w_high <= p.W - 1;
                const bool o3 = h_high <= p.H - 1 && w_low >= 0, o4 = h_high <= p.H - 1 && w_high <= p.W - 1;
                const long a1 = (long)h_low * h_stride + (long)w_low * w_stride, a2 = a1 + w_stride, a3 = a1 + h_stride, a4 = a3 + w_stride;
                for (int c = lane; c < p.D; c += 64) {
                    const A top_grad = (A)gout[c], tg = top_grad * m;
                    A v1 = 0, v2 = 0, v3 = 0, v4 = 0, ghc = 0, gwc = 0;
                    if (o1) { v1 = (A)im[a1 + c]; ghc -= hw * v1; gwc -= hh * v1; atomicAdd(gim + a1 + c, w1 * tg); }
                    if (o2) { v2 = (A)im[a2 + c]; ghc -= lw * v2; gwc += hh * v2; atomicAdd(gim + a2 + c, w2 * tg); }
                    if (o3) { v3 = (A)im[a3 + c]; ghc += hw * v3; gwc -= lh * v3; atomicAdd(gim + a3 + c, w3 * tg); }
                    if (o4) { v4 = (A)im[a4 + c]; ghc += lw * v4; gwc += lh * v4; atomicAdd(gim + a4 + c, w4 * tg); }
                    gm += top_grad * (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
                    gw += os * gwc * tg;
                    gh += os * ghc * tg;
                }
                gm = wave_sum(gm); gw = wave_sum(gw); gh = wave_sum(gh);
                if (lane == 0) { gmask[wptr] = gm; goff[lptr] = gw; goff[lptr + 1] = gh; }
            }
            wptr += 1; lptr += 2;
        }
}

int check_geom(const char* who, const void* a, const void* b, const void* c, const void* d, int N, int H, int W, int G, int D,
               int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int rc, int im2col_step, int dtype, AnyKP& p) {
    GP_REQUIRE(a && b && c && d, "%s: null pointer", who);
    GP_REQUIRE(dtype == GP_F32 || dtype == GP_F16 || dtype == GP_F64, "%s: bad dtype %d", who, dtype);
    GP_REQUIRE(N > 0 && H > 0 && W > 0 && G > 0 && D > 0, "%s: bad shape", who);
    GP_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && dh > 0 && dw > 0, "%s: bad geometry", who);
    GP_REQUIRE(!rc || (kh == kw && (kh & 1)), "%s: remove_center is only compatible with square odd kernel size", who);   // dcnv3_func.py:181-182
    GP_REQUIRE(im2col_step > 0, "%s: im2col_step must be positive", who);
    const int step = N < im2col_step ? N : im2col_step;
    GP_REQUIRE(N % step == 0, "%s: batch(%d) must divide im2col_step(%d)", who, N, step);      // dcnv3_cuda.cu:48-49
    p.N = N; p.H = H; p.W = W; p.G = G; p.D = D; p.kh = kh; p.kw = kw; p.sh = sh; p.sw = sw; p.ph = ph; p.pw = pw; p.dh = dh; p.dw = dw;
    p.rc = rc ? 1 : 0;
    p.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
    p.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
    GP_REQUIRE(p.Ho > 0 && p.Wo > 0, "%s: empty output", who);
    p.rows = (long)N * p.Ho * p.Wo;
    GP_REQUIRE((long)H * W * G * D < (1l << 31), "%s: one image exceeds 2^31 elements", who);
    return GP_OK;
}

}  // namespace

extern "C" int gp_dcnv3_forward_any(const void* in, const void* offset, const void* mask, void* out, int N, int H, int W, int G, int D,
                                    int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float offset_scale,
                                    int remove_center, int im2col_step, int dtype, void* stream) {
    AnyKP p;
    memset(&p, 0, sizeof(p));
    const int rc = check_geom("gp_dcnv3_forward_any", in, offset, mask, out, N, H, W, G, D, kh, kw, sh, sw, ph, pw, dh, dw, remove_center, im2col_step, dtype, p);
    if (rc != GP_OK) return rc;
    p.in = in; p.off = offset; p.mask = mask; p.out = out; p.os = (double)offset_scale;
    hipStream_t s = (hipStream_t)stream;
    const long total = p.rows * G * D, waves = p.rows * G;
    const int P = kh * kw - p.rc, esz = dtype == GP_F16 ? 2 : dtype == GP_F32 ? 4 : 8;
    gp_timing_before(s, GP_KC_DCNV3, (double)total * P * 8.0, ((double)N * H * W * G * D + (double)p.rows * G * P * 3 + (double)total) * esz);
    gp_timing_label("dcnv3_any fwd N%d %dx%d k%dx%d G%d D%d dt%d", N, H, W, kh, kw, G, D, dtype);
    const long nblk = (waves + 3) / 4;   // four wavefronts per workgroup
    GP_REQUIRE(nblk < (1l << 31), "gp_dcnv3_forward_any: grid too large");
    const dim3 grid((unsigned)nblk);
    if (dtype == GP_F16) hipLaunchKernelGGL((dcnv3_any_fwd_kernel<half_t, float>), grid, dim3(256), 0, s, p);
    else if (dtype == GP_F32) hipLaunchKernelGGL((dcnv3_any_fwd_kernel<float, float>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((dcnv3_any_fwd_kernel<double, double>), grid, dim3(256), 0, s, p);
    GP_LAUNCH_CHECK("gp_dcnv3_forward_any");
}

extern "C" int gp_dcnv3_backward(const void* in, const void* offset, const void* mask, const void* grad_out, void* grad_in,
                                 void* grad_offset, void* grad_mask, long offset_numel, long mask_numel, int N, int H, int W, int G,
                                 int D, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, float offset_scale,
                                 int remove_center, int im2col_step, int dtype, void* stream) {
    AnyKP p;
    memset(&p, 0, sizeof(p));
    const int rc = check_geom("gp_dcnv3_backward", in, offset, mask, grad_out, N, H, W, G, D, kh, kw, sh, sw, ph, pw, dh, dw, remove_center, im2col_step, dtype, p);
    if (rc != GP_OK) return rc;
    GP_REQUIRE(grad_in && grad_offset && grad_mask, "gp_dcnv3_backward: null gradient buffer");
    const int P = kh * kw - p.rc;
    GP_REQUIRE(offset_numel >= p.rows * G * P * 2 && mask_numel >= p.rows * G * P, "gp_dcnv3_backward: offset/mask buffers smaller than the kernel consumes");
    p.in = in; p.off = offset; p.mask = mask; p.gout = grad_out; p.gin = grad_in; p.goff = grad_offset; p.gmask = grad_mask;
    p.os = (double)offset_scale;
    hipStream_t s = (hipStream_t)stream;
    const size_t asz = dtype == GP_F64 ? 8 : 4;      // opmath: float for half / float, double for double
    hipError_t e = hipMemsetAsync(grad_in, 0, (size_t)N * H * W * G * D * asz, s);
    if (e == hipSuccess) e = hipMemsetAsync(grad_offset, 0, (size_t)offset_numel * asz, s);
    if (e == hipSuccess) e = hipMemsetAsync(grad_mask, 0, (size_t)mask_numel * asz, s);
    if (e != hipSuccess) return gp_fail(GP_ERR_RUNTIME, "gp_dcnv3_backward: hipMemsetAsync: %s", hipGetErrorString(e));
    const long waves = p.rows * G;
    gp_timing_before(s, GP_KC_DCNV3, (double)waves * D * P * 30.0, ((double)N * H * W * G * D * 2 + (double)waves * P * 6 + (double)waves * D) * asz);
    gp_timing_label("dcnv3 bwd N%d %dx%d k%dx%d G%d D%d dt%d", N, H, W, kh, kw, G, D, dtype);
    const dim3 grid((unsigned)cdiv(waves * 64, 256));
    if (dtype == GP_F16) hipLaunchKernelGGL((dcnv3_any_bwd_kernel<half_t, float>), grid, dim3(256), 0, s, p);
    else if (dtype == GP_F32) hipLaunchKernelGGL((dcnv3_any_bwd_kernel<float, float>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((dcnv3_any_bwd_kernel<double, double>), grid, dim3(256), 0, s, p);
    GP_LAUNCH_CHECK("gp_dcnv3_backward");
}
