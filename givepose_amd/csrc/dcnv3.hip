// DCNv3 forward (deformable bilinear gather) for gfx950.
//
// Follows the arithmetic of the reference CUDA kernel
//   network/ops_dcnv3/src/cuda/dcnv3_im2col_cuda.cuh:216-282 (+ :32-80 bilinear)
// -- same flat offset/mask addressing, tap order (kernel_w outer, kernel_h inner), bounds test,
// zero padding and fp32 accumulation -- but not its thread mapping (one thread per output scalar,
// every one of the D channel threads re-reading the same 27 offset/mask scalars).
//
// Two kernels:
//  * dcnv3_wave_kernel (G*D == 256, D == 64, K*K-rc <= 16: the PoseNet geometry): one wavefront
//    per output pixel, 16 lanes per group, 4 channels per lane.  Lane t < P of each 16-lane row
//    loads tap t's (offset_w, offset_h, mask) once; the softmax over the taps (when the caller hands
//    over mask logits) is a 16-lane butterfly; per tap the three scalars are broadcast inside the
//    row with wave shuffles (no memory traffic) and every corner fetch is one fully coalesced
//    128-B (f16) / 256-B (f32) row segment per group.
//  * dcnv3_generic_kernel: any G, D % 4 == 0, any K: one thread per (pixel, group, 4 channels).
#include "common.hpp"

namespace {

struct DcnKP {
    const void* in;
    const void* off;
    const void* mask;
    void* out;
    int N, H, W, G, D, K, stride, pad, dil, rc, Ho, Wo, off_ld, mask_ld, logits, xcd;
    float os;
    long rows;  // N*Ho*Wo
};

template <typename T> struct Ch4;  // 4 channels of T
template <> struct Ch4<half_t> {
    half4 v;
    __device__ __forceinline__ float get(int i) const { return (float)v[i]; }
    // fma(a, v[i], c): one v_fma_mix_f32 (the fp16 operand is read straight from its half of the packed register; the compiler finds it by itself)
    __device__ __forceinline__ float fma(float a, int i, float c) const { return fmaf(a, (float)v[i], c); }
    __device__ __forceinline__ float mul(float a, int i) const { return a * (float)v[i]; }
};
template <> struct Ch4<float> {
    f32x4 v;
    __device__ __forceinline__ float get(int i) const { return v[i]; }
    __device__ __forceinline__ float mul(float a, int i) const { return a * v[i]; }
    __device__ __forceinline__ float fma(float a, int i, float c) const { return fmaf(a, v[i], c); }
};
template <typename T> __device__ __forceinline__ Ch4<T> ld4(const T* p) {
    Ch4<T> c;
    c.v = *reinterpret_cast<const decltype(c.v)*>(p);
    return c;
}
__device__ __forceinline__ void st4(half_t* p, const float* a) {
    half4 v;
    for (int i = 0; i < 4; ++i) v[i] = (half_t)a[i];
    *reinterpret_cast<half4*>(p) = v;
}
__device__ __forceinline__ void st4(float* p, const float* a) {
    *reinterpret_cast<f32x4*>(p) = f32x4{a[0], a[1], a[2], a[3]};
}

// bilinear sample of 4 channels at (h, w) with zero padding; `im` points at channel 0 of this
// lane's 4 channels of pixel (0,0) of image b; cstride = G*D
template <typename T>
__device__ __forceinline__ void bilinear4(const T* im, int H, int W, int cstride, float h, float w, float wgt,
                                          float* acc) {
    const int h_low = (int)floorf(h), w_low = (int)floorf(w);
    const int h_high = h_low + 1, w_high = w_low + 1;
    const float lh = h - h_low, lw = w - w_low, hh = 1.f - lh, hw = 1.f - lw;
    const bool hl = h_low >= 0, hhi = h_high <= H - 1, wl = w_low >= 0, whi = w_high <= W - 1;
    const long rs = (long)W * cstride;
    const T* p1 = im + h_low * rs + (long)w_low * cstride;
    Ch4<T> v1, v2, v3, v4;
    const bool o1 = hl && wl, o2 = hl && whi, o3 = hhi && wl, o4 = hhi && whi;
    if (o1) v1 = ld4(p1);
    if (o2) v2 = ld4(p1 + cstride);
    if (o3) v3 = ld4(p1 + rs);
    if (o4) v4 = ld4(p1 + rs + cstride);
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float a = o1 ? v1.get(c) : 0.f, b = o2 ? v2.get(c) : 0.f, d = o3 ? v3.get(c) : 0.f,
                    e = o4 ? v4.get(c) : 0.f;
        acc[c] += (w1 * a + w2 * b + w3 * d + w4 * e) * wgt;
    }
}

// Branch-free form for the wave kernel: the four corner fetches are issued unconditionally from CLAMPED coordinates and
// an out-of-range corner gets weight zero (same sum as the reference, which skips it: dcnv3_im2col_cuda.cuh:57-78);
// the products are written as fma(w, (float)half, acc) so that hipcc emits v_fma_mix_f32 (fp16 operand converted inside
// the FMA) instead of a v_cvt_f32_f16 per corner value, and there is no branch between the loads of consecutive taps.
template <typename T>
__device__ __forceinline__ void bilinear4_nb(const T* im, unsigned lane4, int H, int W, float Hf, float Wf, float h, float w, float wgt, float* acc) {
    const float fh = floorf(h), fw = floorf(w);
    const float lh = h - fh, lw = w - fw, hh = 1.f - lh, hw = 1.f - lw;
    const int h_low = (int)fh, w_low = (int)fw;          // in (-2, H) x (-2, W): the caller's range test guarantees it
    const int h_high = h_low + 1, w_high = w_low + 1;
    const bool hl = h_low >= 0, hhi = h_high <= H - 1, wl = w_low >= 0, whi = w_high <= W - 1;
    const float w1 = (hl && wl) ? hh * hw : 0.f, w2 = (hl && whi) ? hh * lw : 0.f;
    const float w3 = (hhi && wl) ? lh * hw : 0.f, w4 = (hhi && whi) ? lh * lw : 0.f;
    const int y0 = max(h_low, 0), y1 = min(h_high, H - 1), x0 = max(w_low, 0), x1 = min(w_high, W - 1);
    // 32-bit byte offsets from the (wave-uniform) image base: scalar-base loads, no 64-bit vector address arithmetic
    const unsigned r0 = (unsigned)(y0 * W) * 256u + lane4, r1 = (unsigned)(y1 * W) * 256u + lane4;
    const char* base = reinterpret_cast<const char*>(im);
    const Ch4<T> v1 = ld4(reinterpret_cast<const T*>(base + (size_t)((r0 + x0 * 256u) * (unsigned)sizeof(T)))),
                 v2 = ld4(reinterpret_cast<const T*>(base + (size_t)((r0 + x1 * 256u) * (unsigned)sizeof(T)))),
                 v3 = ld4(reinterpret_cast<const T*>(base + (size_t)((r1 + x0 * 256u) * (unsigned)sizeof(T)))),
                 v4 = ld4(reinterpret_cast<const T*>(base + (size_t)((r1 + x1 * 256u) * (unsigned)sizeof(T))));
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float t = w1 * v1.get(c);
        t = fmaf(w2, v2.get(c), t);
        t = fmaf(w3, v3.get(c), t);
        t = fmaf(w4, v4.get(c), t);
        acc[c] = fmaf(t, wgt, acc[c]);
    }
}

template <typename T, typename OT>
__global__ __launch_bounds__(256) void dcnv3_generic_kernel(const DcnKP p) {
    const int DV = p.D >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.rows * p.G * DV) return;
    const int cv = (int)(idx % DV);
    long t = idx / DV;
    const int g = (int)(t % p.G);
    const long r = t / p.G;
    const int wo = (int)(r % p.Wo);
    const long t2 = r / p.Wo;
    const int ho = (int)(t2 % p.Ho);
    const int b = (int)(t2 / p.Ho);
    const int P = p.K * p.K - p.rc;
    const OT* op = reinterpret_cast<const OT*>(p.off) + r * p.off_ld + g * P * 2;
    const OT* mp = reinterpret_cast<const OT*>(p.mask) + r * p.mask_ld + g * P;
    float mmax = 0.f, minv = 1.f;
    if (p.logits) {
        mmax = -INFINITY;
        for (int q = 0; q < P; ++q) mmax = fmaxf(mmax, (float)mp[q]);
        float s = 0.f;
        for (int q = 0; q < P; ++q) s += expf((float)mp[q] - mmax);
        minv = 1.f / s;
    }
    const int halfk = (p.dil * (p.K - 1)) >> 1;
    const float p0_w_ = (float)(halfk - p.pad + wo * p.stride) - halfk * p.os;
    const float p0_h_ = (float)(halfk - p.pad + ho * p.stride) - halfk * p.os;
    const int cs = p.G * p.D;
    const T* im = reinterpret_cast<const T*>(p.in) + (long)b * p.H * p.W * cs + g * p.D + cv * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int q = 0;
    for (int i = 0; i < p.K; ++i)
        for (int j = 0; j < p.K; ++j) {
            if (p.rc && i == p.K / 2 && j == p.K / 2) continue;
            const float ow = (float)op[2 * q], oh = (float)op[2 * q + 1];
            const float loc_w = p0_w_ + (i * p.dil + ow) * p.os;
            const float loc_h = p0_h_ + (j * p.dil + oh) * p.os;
            float wgt = (float)mp[q];
            if (p.logits) wgt = expf(wgt - mmax) * minv;
            if (loc_h > -1.f && loc_w > -1.f && loc_h < (float)p.H && loc_w < (float)p.W)
                bilinear4<T>(im, p.H, p.W, cs, loc_h, loc_w, wgt, acc);
            ++q;
        }
    st4(reinterpret_cast<T*>(p.out) + r * cs + g * p.D + cv * 4, acc);
}

// One wavefront per output pixel; G == 4, D == 64 (lane = g*16 + cv), P <= 16.
// KS > 0: kernel size known at compile time and remove_center == 0 (the PoseNet geometry is KS = 3): the tap loops
// unroll and the (tap index -> i, j) bookkeeping disappears; the arithmetic per tap is the same expression.
// PATCH (KS == 3, Ho % 4 == 0, Wo % 4 == 0): a workgroup is a 4 x 4 patch of output pixels of ONE group (blockIdx.y): wave =
// patch row, 16-lane row = pixel, lane = 4 of the group's 64 channels.  The taps of neighbouring output pixels land on
// overlapping input pixels, and one group's share of an input pixel is a single 128-byte line, so the patch's footprint
// (about 10 x 10 input pixels = 13 KB) stays in the CU's 32-KB L1: the wave-per-pixel mapping below pulls every one of its
// 36 x 512 gathered bytes from L2 (1.2 GB per launch at 64 x 64 = 4.7 MB per CU, i.e. the ~55 GB/s a CU gets from L2).
template <typename T, typename OT, int KS = 0, bool PATCH = false, bool FOLD = false>     // FOLD = true: measurement arm (GP_DCN_FOLD=1), see MW below
__global__ __launch_bounds__(256) void dcnv3_wave_kernel(const DcnKP p) {
    static_assert(!PATCH || KS == 3, "patch mapping: 3 x 3 only");
    const int lane = threadIdx.x & 63;
    long r;
    int g, wo, ho, b;
    const int t = lane & 15;
    if constexpr (PATCH) {
        const int ppr = p.Wo >> 2, ppi = ppr * (p.Ho >> 2);
        // XCD x (workgroup ids equal mod 8 share one L2) owns a CONTIGUOUS run of patches: neighbouring patches read overlapping
        // input footprints (stride 2, 3x3 taps, offsets of a few pixels: ~16 x 16 input pixels for 8 x 8 "owned" ones), and with
        // the round-robin order each of them sat in a different L2 -- the halos then came from HBM / Infinity Cache once per XCD
        // (PMC: 1.72 x the algorithmic bytes in round 2).  GP_DCN_XCD=0 keeps the old order (A/B).
        const int pid = p.xcd ? xcd_chunk(blockIdx.x, gridDim.x) : blockIdx.x;
        b = pid / ppi;
        const int pin = pid - b * ppi;
        ho = (pin / ppr) * 4 + (threadIdx.x >> 6);
        wo = (pin % ppr) * 4 + (lane >> 4);
        g = blockIdx.y;
        r = ((long)b * p.Ho + ho) * p.Wo + wo;
    } else {
        r = (long)(p.xcd ? xcd_chunk(blockIdx.x, gridDim.x) : blockIdx.x) * 4 + (threadIdx.x >> 6);
        if (r >= p.rows) return;  // whole wave exits together
        g = lane >> 4;
        wo = (int)(r % p.Wo);
        const long t2 = r / p.Wo;
        ho = (int)(t2 % p.Ho);
        b = (int)(t2 / p.Ho);
    }
    const int K = KS > 0 ? KS : p.K, rc = KS > 0 ? 0 : p.rc;
    const int P = K * K - rc;
    // lane t < P of row g owns tap t
    float ow = 0.f, oh = 0.f, mk = p.logits ? -INFINITY : 0.f;
    if (t < P) {
        const OT* op = reinterpret_cast<const OT*>(p.off) + r * p.off_ld + (g * P + t) * 2;
        ow = (float)op[0];
        oh = (float)op[1];
        mk = (float)(reinterpret_cast<const OT*>(p.mask)[r * p.mask_ld + g * P + t]);
    }
    if (p.logits) {
        const float mx = group_max(mk, 16);
        const float e = t < P ? expf(mk - mx) : 0.f;
        const float s = group_sum(e, 16);
        mk = e / s;
    }
    const int halfk = (p.dil * (K - 1)) >> 1;
    const float p0_w_ = (float)(halfk - p.pad + wo * p.stride) - halfk * p.os;
    const float p0_h_ = (float)(halfk - p.pad + ho * p.stride) - halfk * p.os;
    const int cl = g * 16 + t;   // this lane's channel quad (== lane in the wave-per-pixel mapping)
    const T* im = reinterpret_cast<const T*>(p.in) + (long)b * p.H * p.W * 256 + cl * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (KS > 0) {
        // Lane t < 9 of every 16-lane row owns tap t of its group and works out ONCE what all 16 lanes of the row need
        // for that tap -- corner weights (zero for an out-of-range corner or tap), the mask weight, the byte offset of
        // the top-left corner and the steps to the right / lower corner -- ; the tap loop then only broadcasts those
        // eight numbers (DPP row_share, no LDS crossbar) and does the four loads + 20 FMAs.  The first version
        // repeated the ~70 instructions of coordinate arithmetic per tap in every lane: 1 100 VALU instructions per
        // output pixel, which is what bounded the kernel (ablation: 58 of 120 us with every load removed).
        static_assert(KS == 3, "row_share immediates are spelled out for 3x3");
        constexpr bool MW = sizeof(T) == 2 && FOLD;
        const float Hf = (float)p.H, Wf = (float)p.W;
        const char* imb = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.in) +
                                                        (long)__builtin_amdgcn_readfirstlane(b) * p.H * p.W * 256);
        float w1, w2, w3, w4;
        unsigned o00, dx, dy;
        {
            const int i = t / 3, j = t - i * 3;            // tap order: kernel_w outer, kernel_h inner
            float loc_w = p0_w_ + (i * p.dil + ow) * p.os;
            float loc_h = p0_h_ + (j * p.dil + oh) * p.os;
            const bool in = t < 9 && loc_h > -1.f && loc_w > -1.f && loc_h < Hf && loc_w < Wf;
            loc_w = in ? loc_w : 0.f;
            loc_h = in ? loc_h : 0.f;
            mk = in ? mk : 0.f;
            const float fh = floorf(loc_h), fw = floorf(loc_w);
            const float lh = loc_h - fh, lw = loc_w - fw, hh = 1.f - lh, hw = 1.f - lw;
            const int h_low = (int)fh, w_low = (int)fw, h_high = h_low + 1, w_high = w_low + 1;
            const bool hl = h_low >= 0, hhi = h_high <= p.H - 1, wl = w_low >= 0, whi = w_high <= p.W - 1;
            w1 = (hl && wl) ? hh * hw : 0.f;
            w2 = (hl && whi) ? hh * lw : 0.f;
            w3 = (hhi && wl) ? lh * hw : 0.f;
            w4 = (hhi && whi) ? lh * lw : 0.f;
            const int y0 = max(h_low, 0), y1 = min(h_high, p.H - 1), x0 = max(w_low, 0), x1 = min(w_high, p.W - 1);
            if constexpr (MW) { w1 *= mk; w2 *= mk; w3 *= mk; w4 *= mk; }
            o00 = (unsigned)((y0 * p.W + x0) * 256) * (unsigned)sizeof(T);
            dx = (unsigned)((x1 - x0) * 256) * (unsigned)sizeof(T);
            dy = (unsigned)((y1 - y0) * p.W * 256) * (unsigned)sizeof(T);
        }
        const unsigned lo = (unsigned)cl * 4u * (unsigned)sizeof(T);
        // MW (fp16 storage, GP_DCN_FOLD=1 only): the mask weight folded into the four corner weights by the tap's owner lane (4 multiplies per tap and ROW
        // instead of one FMA per tap, channel and LANE, and one broadcast less: 433 -> 392 VALU instructions per wave).  Measured: -2 % at 64 x 64, nothing
        // below (profiles/r06_dcn_fold_ab.txt) -- the gather is bound by its line requests -- and another rounding of every gathered value, which the fp16
        // mode's worst-conditioned crop can see (tests/test_grouped_launch.py: |dR| 2.0e-2 -> 3.8e-2 against a 3e-2 bound).  Not the default: every
        // instantiation keeps the reference's association ((w1 v1 + w2 v2 + w3 v3 + w4 v4) * mask, dcnv3_im2col_cuda.cuh:32-80).
#define GP_RSF(v, q) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + (q), 0xF, 0xF, true))
#define GP_RSU(v, q) (unsigned)__builtin_amdgcn_update_dpp(0, (int)(v), 0x150 + (q), 0xF, 0xF, true)
#define GP_TAP(q)                                                                                              \
        {                                                                                                      \
            const float a1 = GP_RSF(w1, q), a2 = GP_RSF(w2, q), a3 = GP_RSF(w3, q), a4 = GP_RSF(w4, q), wg = MW ? 0.f : GP_RSF(mk, q); \
            const unsigned b00 = GP_RSU(o00, q) + lo, ex = GP_RSU(dx, q), ey = GP_RSU(dy, q);                  \
            const Ch4<T> v1 = ld4(reinterpret_cast<const T*>(imb + (size_t)b00)),                              \
                         v2 = ld4(reinterpret_cast<const T*>(imb + (size_t)(b00 + ex))),                       \
                         v3 = ld4(reinterpret_cast<const T*>(imb + (size_t)(b00 + ey))),                       \
                         v4 = ld4(reinterpret_cast<const T*>(imb + (size_t)(b00 + ex + ey)));                  \
            if constexpr (MW) {                                                                                \
                _Pragma("unroll") for (int c = 0; c < 4; ++c)                                                  \
                    acc[c] = v4.fma(a4, c, v3.fma(a3, c, v2.fma(a2, c, v1.fma(a1, c, acc[c]))));               \
            } else                                                                                             \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                    \
                float tt = v1.mul(a1, c);                                                                      \
                tt = fmaf(a2, v2.get(c), tt);                                                                  \
                tt = fmaf(a3, v3.get(c), tt);                                                                  \
                tt = fmaf(a4, v4.get(c), tt);                                                                  \
                acc[c] = fmaf(tt, wg, acc[c]);                                                                 \
            }                                                                                                  \
        }
        GP_TAP(0) GP_TAP(1) GP_TAP(2) GP_TAP(3) GP_TAP(4) GP_TAP(5) GP_TAP(6) GP_TAP(7) GP_TAP(8)
#undef GP_TAP
#undef GP_RSU
#undef GP_RSF
    } else {
        int q = 0;
        for (int i = 0; i < p.K; ++i)
            for (int j = 0; j < p.K; ++j) {
                if (p.rc && i == p.K / 2 && j == p.K / 2) continue;
                const int src = (lane & 48) | q;  // tap q's owner inside this 16-lane row
                const float tw = __shfl(ow, src, 64), th = __shfl(oh, src, 64), wgt = __shfl(mk, src, 64);
                const float loc_w = p0_w_ + (i * p.dil + tw) * p.os;
                const float loc_h = p0_h_ + (j * p.dil + th) * p.os;
                if (loc_h > -1.f && loc_w > -1.f && loc_h < (float)p.H && loc_w < (float)p.W)
                    bilinear4<T>(im, p.H, p.W, 256, loc_h, loc_w, wgt, acc);
                ++q;
            }
    }
    st4(reinterpret_cast<T*>(p.out) + r * 256 + cl * 4, acc);
}

// ---------------------------------------------------------------------------- fp16, 3 x 3, SIXTEEN bytes per lane (round 6)
// Why: a wave-level load instruction costs the CU's vector-memory path the same ~18 cycles at 8 and at 16 bytes per lane
// (scripts/probes/gather_width.hip, profiles/r06_gather_width.txt: 28 against 50-57 B/clk/CU out of L1 / L2), and the kernel above -- 4 fp16 channels
// = 8 bytes per lane, 36 loads per wave for 4 (pixel, group) pairs -- ran at exactly that rate whatever the offsets and the occupancy (80 us at 64 x 64,
// offsets 0 .. 8 pixels, 3 .. 8 workgroups per CU: profiles/r06_dcn_wave8_ab.txt).  Here a lane owns EIGHT channels: a 16-lane row is one output
// pixel x TWO groups (lanes 0-7 group 2 gp, lanes 8-15 group 2 gp + 1), a wave 4 pixels x 2 groups, so the same 36 loads serve 8 pairs.  Lanes t < 9
// of a row own tap t of BOTH groups (the owner block of the kernel above, run once per group); the tap loop hands lanes 0-7 the first group's eight
// numbers and lanes 8-15 the second's: two DPP row_share moves per number with bank masks 0x3 / 0xC.  Per channel the arithmetic and its order are
// those of the kernel above: the outputs are bitwise the same (tests/test_hip_ops.py).
typedef unsigned pk_u32x4 __attribute__((ext_vector_type(4)));
// a * (fp16 half `hi` of u) in fp32 as ONE instruction: v_fma_mix_f32 with a zero addend reads the fp16 operand straight from its half of the packed register
// (the compiler's cvt + mul are two).  The bits of the product, except that a zero product comes out as +0 -- which no sum downstream can tell
// (the accumulators start at +0): checked bitwise against dcnv3_wave_kernel, where most corner weights of an edge pixel ARE zero.
__device__ __forceinline__ float mul_mix(float a, unsigned u, int hi) {
    float r;
    if (hi) asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(u));
    else asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(u));
    return r;
}

// LB (measurement arm, GP_DCN_LDSBC=1): the eight numbers of a (pixel, group, tap) go from their owner lane to the 8 lanes that need them through LDS (two 16-byte
// broadcast reads per tap, issued a stage ahead) instead of 16 bank-masked DPP moves per tap: 771 -> ~640 vector instructions per wave, same bits -- and the same
// 52 us at 64 x 64 (profiles/r06_dcn_wave8_ab.txt [5]): with 16-byte loads the kernel sits on its load instructions again (36 per wave x ~18 cycles = 35 us at 2.4 GHz),
// not on vector issue.  Not the default.
template <typename OT, bool LB>
__global__ __launch_bounds__(256) void dcnv3_wave8_kernel(const DcnKP p) {
    __shared__ __attribute__((aligned(16))) uint4 bc_s[LB ? 4 * 72 * 2 : 1];       // [wave][(pixel row of the wave, group of the pair, tap)][weights | mask, offsets]
    const int lane = threadIdx.x & 63, t = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ppr = p.Wo >> 2, ppi = ppr * (p.Ho >> 2);
    const int pid = p.xcd ? xcd_chunk(blockIdx.x, gridDim.x) : blockIdx.x;
    const int b = pid / ppi, pin = pid - b * ppi;
    const int ho = (pin / ppr) * 4 + (threadIdx.x >> 6), wo = (pin % ppr) * 4 + (lane >> 4);
    const long r = ((long)b * p.Ho + ho) * p.Wo + wo;
    const int halfk = (p.dil * 2) >> 1;
    const float p0_w_ = (float)(halfk - p.pad + wo * p.stride) - halfk * p.os;
    const float p0_h_ = (float)(halfk - p.pad + ho * p.stride) - halfk * p.os;
    const float Hf = (float)p.H, Wf = (float)p.W;
    const char* imb = reinterpret_cast<const char*>(reinterpret_cast<const half_t*>(p.in) + (long)__builtin_amdgcn_readfirstlane(b) * p.H * p.W * 256);
    float w1[2], w2[2], w3[2], w4[2], mk[2];
    unsigned o00[2], dx[2], dy[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {          // owner block (as dcnv3_wave_kernel), once per group of the pair
        const int g = blockIdx.y * 2 + h;
        float ow = 0.f, oh = 0.f, m = p.logits ? -INFINITY : 0.f;
        if (t < 9) {
            const OT* op = reinterpret_cast<const OT*>(p.off) + r * p.off_ld + (g * 9 + t) * 2;
            ow = (float)op[0];
            oh = (float)op[1];
            m = (float)(reinterpret_cast<const OT*>(p.mask)[r * p.mask_ld + g * 9 + t]);
        }
        if (p.logits) {
            const float mx = group_max(m, 16);
            const float e = t < 9 ? expf(m - mx) : 0.f;
            const float sm = group_sum(e, 16);
            m = e / sm;
        }
        const int i = t / 3, j = t - i * 3;            // tap order: kernel_w outer, kernel_h inner
        float loc_w = p0_w_ + (i * p.dil + ow) * p.os;
        float loc_h = p0_h_ + (j * p.dil + oh) * p.os;
        const bool in = t < 9 && loc_h > -1.f && loc_w > -1.f && loc_h < Hf && loc_w < Wf;
        loc_w = in ? loc_w : 0.f;
        loc_h = in ? loc_h : 0.f;
        m = in ? m : 0.f;
        const float fh = floorf(loc_h), fw = floorf(loc_w);
        const float lh = loc_h - fh, lw = loc_w - fw, hh = 1.f - lh, hw = 1.f - lw;
        const int h_low = (int)fh, w_low = (int)fw, h_high = h_low + 1, w_high = w_low + 1;
        const bool hl = h_low >= 0, hhi = h_high <= p.H - 1, wl = w_low >= 0, whi = w_high <= p.W - 1;
        w1[h] = (hl && wl) ? hh * hw : 0.f;
        w2[h] = (hl && whi) ? hh * lw : 0.f;
        w3[h] = (hhi && wl) ? lh * hw : 0.f;
        w4[h] = (hhi && whi) ? lh * lw : 0.f;
        mk[h] = m;
        const int y0 = max(h_low, 0), y1 = min(h_high, p.H - 1), x0 = max(w_low, 0), x1 = min(w_high, p.W - 1);
        o00[h] = (unsigned)((y0 * p.W + x0) * 256) * 2u;
        dx[h] = (unsigned)((x1 - x0) * 256) * 2u;
        dy[h] = (unsigned)((y1 - y0) * p.W * 256) * 2u;
        if constexpr (LB) {
            if (t < 9) {
                uint4* rec = bc_s + ((wave * 72 + ((lane >> 4) * 2 + h) * 9 + t) * 2);
                rec[0] = uint4{__builtin_bit_cast(unsigned, w1[h]), __builtin_bit_cast(unsigned, w2[h]), __builtin_bit_cast(unsigned, w3[h]), __builtin_bit_cast(unsigned, w4[h])};
                rec[1] = uint4{__builtin_bit_cast(unsigned, m), o00[h], dx[h], dy[h]};
            }
        }
    }
    if constexpr (LB) __builtin_amdgcn_wave_barrier();      // (a wave's LDS instructions execute in order: the reads below see these writes; this only pins the compiler's order)
    const uint4* myrec = bc_s + (wave * 72 + ((lane >> 4) * 2 + (t >> 3)) * 9) * 2;
    const unsigned lo = (unsigned)(blockIdx.y * 128 + t * 8) * 2u;      // this lane's 8 channels inside a pixel (bytes)
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    // Software pipeline, spelled out: the four loads of tap q + 2 are issued before tap q is multiplied (12 loads = 12 KB per wave in flight), with
    // scheduling barriers between the stages -- left alone, the compiler hoisted loads until it needed 159 registers (3 waves per SIMD: 77 us at 64 x 64
    // against 57) and spilled when held to 128.
    int a1 = 0, a2 = 0, a3 = 0, a4 = 0, wg = 0, bo = 0, ex = 0, ey = 0;          // broadcast destinations (kept across taps: no re-initialisation)
    half8 v[3][4];
    uint4 ra[3];          // LB: the four corner weights of the stage's tap
    unsigned rg[3];       // LB: its mask weight
    // lanes 0-7 of a row <- lane q's value for the first group (bank mask 0x3), lanes 8-15 <- lane q's value for the second (0xC); the first halves of a
    // stage's numbers are issued together, then the second halves (a DPP move may not read a register the instruction before it wrote)
#define GP_BCA(d, A, q) d = __builtin_amdgcn_update_dpp(d, __builtin_bit_cast(int, A), 0x150 + (q), 0xF, 0x3, false);
#define GP_BCB(d, B, q) d = __builtin_amdgcn_update_dpp(d, __builtin_bit_cast(int, B), 0x150 + (q), 0xF, 0xC, false);
    auto fetch = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (LB) {
            const uint4 rb = myrec[q * 2 + 1];
            ra[q % 3] = myrec[q * 2];
            rg[q % 3] = rb.x;
            bo = (int)rb.y; ex = (int)rb.z; ey = (int)rb.w;
        } else {
        GP_BCA(bo, o00[0], q) GP_BCA(ex, dx[0], q) GP_BCA(ey, dy[0], q)
        GP_BCB(bo, o00[1], q) GP_BCB(ex, dx[1], q) GP_BCB(ey, dy[1], q)
        }
        const unsigned b00 = (unsigned)bo + lo;
        v[q % 3][0] = *reinterpret_cast<const half8*>(imb + (size_t)b00);
        v[q % 3][1] = *reinterpret_cast<const half8*>(imb + (size_t)(b00 + (unsigned)ex));
        v[q % 3][2] = *reinterpret_cast<const half8*>(imb + (size_t)(b00 + (unsigned)ey));
        v[q % 3][3] = *reinterpret_cast<const half8*>(imb + (size_t)(b00 + (unsigned)ex + (unsigned)ey));
    };
    fetch(std::integral_constant<int, 0>{});
    fetch(std::integral_constant<int, 1>{});
    static_for<0, 9>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q + 2 < 9) fetch(std::integral_constant<int, q + 2>{});
        asm volatile("" ::: "memory");             // (the loads of later taps stay behind this point: a scheduling barrier alone did not hold them)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (LB) {
            a1 = (int)ra[q % 3].x; a2 = (int)ra[q % 3].y; a3 = (int)ra[q % 3].z; a4 = (int)ra[q % 3].w; wg = (int)rg[q % 3];
        } else {
        GP_BCA(a1, w1[0], q) GP_BCA(a2, w2[0], q) GP_BCA(a3, w3[0], q) GP_BCA(a4, w4[0], q) GP_BCA(wg, mk[0], q)
        GP_BCB(a1, w1[1], q) GP_BCB(a2, w2[1], q) GP_BCB(a3, w3[1], q) GP_BCB(a4, w4[1], q) GP_BCB(wg, mk[1], q)
        }
        const float f1 = __builtin_bit_cast(float, a1), f2 = __builtin_bit_cast(float, a2), f3 = __builtin_bit_cast(float, a3),
                    f4 = __builtin_bit_cast(float, a4), fg = __builtin_bit_cast(float, wg);
        const half8 v1 = v[q % 3][0], v2 = v[q % 3][1], v3 = v[q % 3][2], v4 = v[q % 3][3];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float tt = mul_mix(f1, __builtin_bit_cast(pk_u32x4, v1)[c >> 1], c & 1);
            tt = fmaf(f2, (float)v2[c], tt);
            tt = fmaf(f3, (float)v3[c], tt);
            tt = fmaf(f4, (float)v4[c], tt);
            acc[c] = fmaf(tt, fg, acc[c]);
        }
        // (pins the tap's arithmetic here: the compiler otherwise sank the FMAs of taps 1 .. 8 behind the last load, with every loaded value live until then)
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]), "+v"(acc[7]));
        __builtin_amdgcn_sched_barrier(0);
    });
#undef GP_BCA
#undef GP_BCB
    half8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (half_t)acc[c];
    *reinterpret_cast<half8*>(reinterpret_cast<half_t*>(p.out) + r * 256 + blockIdx.y * 128 + t * 8) = o;
}

static bool dcn_ldsbc_enabled() {   // GP_DCN_LDSBC=1: measurement arm, read per call: dcnv3_wave8_kernel with LDS records instead of the DPP broadcasts
    const char* e = getenv("GP_DCN_LDSBC");
    return e && e[0] == '1';
}

static bool dcn_wave8_enabled() {   // GP_DCN_WAVE8=0: A/B switch, read per call (tests flip it inside one process): the 8-bytes-per-lane kernel
    const char* e = getenv("GP_DCN_WAVE8");
    return !(e && e[0] == '0');
}

static bool dcn_xcd_enabled() {   // GP_DCN_XCD=0: A/B switch for the XCD-contiguous workgroup order
    static const bool on = [] { const char* e = getenv("GP_DCN_XCD"); return !(e && e[0] == '0'); }();
    return on;
}

static bool dcn_fold_enabled() {   // GP_DCN_FOLD=1: measurement arm (fp16: mask weight folded into the corner weights; off by default, see MW in the kernel)
    static const bool on = [] { const char* e = getenv("GP_DCN_FOLD"); return e && e[0] == '1'; }();
    return on;
}

static int dcn_lds_pad() {   // GP_DCN_LDS_PAD=<bytes>: unused dynamic LDS per workgroup = fewer workgroups per CU (experiment: L1 footprint)
    static const int k = [] { const char* e = getenv("GP_DCN_LDS_PAD"); return e ? atoi(e) : 0; }();
    return k;
}

static bool dcn_patch_enabled() {   // GP_DCN_PATCH=0: A/B switch
    static const bool on = [] { const char* e = getenv("GP_DCN_PATCH"); return !(e && e[0] == '0'); }();
    return on;
}

template <typename T, typename OT> int launch(const DcnKP& p, hipStream_t s) {
    if constexpr (sizeof(T) == 2) {
        if (p.G == 4 && p.D == 64 && p.K == 3 && !p.rc && p.Ho % 4 == 0 && p.Wo % 4 == 0 && dcn_patch_enabled() && dcn_wave8_enabled() && !dcn_fold_enabled()) {
            if (dcn_ldsbc_enabled()) hipLaunchKernelGGL((dcnv3_wave8_kernel<OT, true>), dim3(p.N * (p.Ho / 4) * (p.Wo / 4), 2), dim3(256), 0, s, p);
            else hipLaunchKernelGGL((dcnv3_wave8_kernel<OT, false>), dim3(p.N * (p.Ho / 4) * (p.Wo / 4), 2), dim3(256), 0, s, p);
            return 0;
        }
    }
    if (p.G == 4 && p.D == 64 && p.K == 3 && !p.rc && p.Ho % 4 == 0 && p.Wo % 4 == 0 && dcn_patch_enabled()) {
        if (sizeof(T) == 2 && dcn_fold_enabled()) hipLaunchKernelGGL((dcnv3_wave_kernel<T, OT, 3, true, true>), dim3(p.N * (p.Ho / 4) * (p.Wo / 4), 4), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((dcnv3_wave_kernel<T, OT, 3, true>), dim3(p.N * (p.Ho / 4) * (p.Wo / 4), 4), dim3(256), dcn_lds_pad(), s, p);
    } else if (p.G == 4 && p.D == 64 && p.K == 3 && !p.rc) {
        hipLaunchKernelGGL((dcnv3_wave_kernel<T, OT, 3>), dim3(cdiv(p.rows, 4)), dim3(256), 0, s, p);
    } else if (p.G == 4 && p.D == 64 && p.K * p.K - p.rc <= 16) {
        hipLaunchKernelGGL((dcnv3_wave_kernel<T, OT>), dim3(cdiv(p.rows, 4)), dim3(256), 0, s, p);
    } else {
        const long total = p.rows * p.G * (p.D / 4);
        hipLaunchKernelGGL((dcnv3_generic_kernel<T, OT>), dim3(cdiv(total, 256)), dim3(256), 0, s, p);
    }
    return 0;
}

}  // namespace

extern "C" int gp_dcnv3_forward(const void* in, const void* offset, const void* mask, void* out, int N, int H,
                                int W, int G, int D, int K, int stride, int pad, int dil, float offset_scale,
                                int remove_center, int im2col_step, int off_ld, int mask_ld,
                                int mask_is_logits, int dtype, int om_dtype, void* stream) {
    GP_REQUIRE(in && offset && mask && out, "gp_dcnv3_forward: null pointer");
    GP_REQUIRE(N > 0 && H > 0 && W > 0 && G > 0 && D > 0 && K > 0 && stride > 0 && dil > 0 && pad >= 0,
               "gp_dcnv3_forward: bad geometry");
    GP_REQUIRE(D % 4 == 0, "gp_dcnv3_forward: group_channels=%d must be a multiple of 4", D);
    GP_REQUIRE((dtype == GP_F32 || dtype == GP_F16) && (om_dtype == GP_F32 || om_dtype == GP_F16),
               "gp_dcnv3_forward: bad dtype");
    GP_REQUIRE(!remove_center || (K % 2 == 1), "remove_center is only compatible with odd kernel size.");
    const int step = im2col_step < N ? im2col_step : N;
    // dcnv3_cuda.cu:46-49
    GP_REQUIRE(step > 0 && N % step == 0, "batch(%d) must divide im2col_step(%d)", N, step);
    DcnKP p;
    p.in = in; p.off = offset; p.mask = mask; p.out = out;
    p.N = N; p.H = H; p.W = W; p.G = G; p.D = D; p.K = K; p.stride = stride; p.pad = pad; p.dil = dil;
    p.rc = remove_center ? 1 : 0;
    p.Ho = (H + 2 * pad - (dil * (K - 1) + 1)) / stride + 1;
    p.Wo = (W + 2 * pad - (dil * (K - 1) + 1)) / stride + 1;
    GP_REQUIRE(p.Ho > 0 && p.Wo > 0, "gp_dcnv3_forward: empty output");
    const int P = K * K - p.rc;
    GP_REQUIRE(off_ld >= G * P * 2 && mask_ld >= G * P, "gp_dcnv3_forward: off_ld/mask_ld too small");
    p.off_ld = off_ld; p.mask_ld = mask_ld; p.logits = mask_is_logits ? 1 : 0; p.os = offset_scale;
    p.xcd = dcn_xcd_enabled() ? 1 : 0;
    p.rows = (long)N * p.Ho * p.Wo;
    hipStream_t s = (hipStream_t)stream;
    const int esz = dtype == GP_F16 ? 2 : 4, osz = om_dtype == GP_F16 ? 2 : 4;
    // algorithmic bytes: input once + consumed offset/mask + output
    const double bytes = (double)N * H * W * G * D * esz + (double)p.rows * G * P * 3 * osz + (double)p.rows * G * D * esz;
    gp_timing_before(s, GP_KC_DCNV3, (double)p.rows * G * D * P * 8.0, bytes);
    gp_timing_label("dcnv3 N%d %dx%d s%d G%d D%d", N, H, W, stride, G, D);
    if (dtype == GP_F16 && om_dtype == GP_F16) launch<half_t, half_t>(p, s);
    else if (dtype == GP_F16) launch<half_t, float>(p, s);
    else if (om_dtype == GP_F16) launch<float, half_t>(p, s);
    else launch<float, float>(p, s);
    GP_LAUNCH_CHECK("gp_dcnv3_forward");
}
