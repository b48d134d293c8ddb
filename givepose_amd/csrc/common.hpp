// Shared helpers for the givepose_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/givepose_hip.h"

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------- errors
extern thread_local char gp_err_buf[512];
int gp_fail(int code, const char* fmt, ...);

#define GP_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return gp_fail(GP_ERR_INVALID, __VA_ARGS__); \
    } while (0)

#define GP_LAUNCH_CHECK(name)                                                              \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) return gp_fail(GP_ERR_LAUNCH, "%s: %s", name, hipGetErrorString(e__)); \
        return gp_timing_after(name);                                                      \
    } while (0)

// per-launch timing hooks (bench.py's roofline leg); no-ops unless gp_timing_begin() was called
void gp_timing_before(hipStream_t s, int cls, double flops, double bytes);
int gp_timing_after(const char* name);
void gp_timing_label(const char* fmt, ...);   // optional, between before/after: groups the launch under this label in gp_timing_top

// ---------------------------------------------------------------------------------- vectors
// One 16-byte vector of T: 8 halfs or 4 floats.
template <typename T> struct Vec16;
template <> struct Vec16<half_t> {
    static constexpr int N = 8;
    union { uint4 u; half_t e[8]; };
    __device__ __forceinline__ float get(int i) const { return (float)e[i]; }
    __device__ __forceinline__ void set(int i, float v) { e[i] = (half_t)v; }
};
template <> struct Vec16<float> {
    static constexpr int N = 4;
    union { uint4 u; float e[4]; };
    __device__ __forceinline__ float get(int i) const { return e[i]; }
    __device__ __forceinline__ void set(int i, float v) { e[i] = v; }
};

template <typename T> __device__ __forceinline__ Vec16<T> load16(const T* p) {
    Vec16<T> v;
    v.u = *reinterpret_cast<const uint4*>(p);
    return v;
}
template <typename T> __device__ __forceinline__ void store16(T* p, const Vec16<T>& v) {
    *reinterpret_cast<uint4*>(p) = v.u;
}
// Split-operand planes as an OUTPUT format of the fp32 kernels that feed a split-operand GEMM (dtype = GP_F32 | GP_OUT_PLANES):
// instead of 4 fp32 at element offset `off` of y, write hi = fp16(v) at half-offset `off` of the same storage and
// lo' = fp16((v - hi) * 2^GP_SPLIT_SHIFT) `pl` halfs behind it (pl = elements of the dense output tensor; 0 = plain store).
template <typename T> __device__ __forceinline__ void store16p(T* y, long off, const Vec16<T>& v, long pl) {
    if constexpr (sizeof(T) == 4) {
        if (pl) {
            half4 h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (half_t)v.e[e];
                l[e] = (half_t)((v.e[e] - (float)h[e]) * (float)(1 << GP_SPLIT_SHIFT));
            }
            half_t* b = reinterpret_cast<half_t*>(y) + off;
            *reinterpret_cast<half4*>(b) = h;
            *reinterpret_cast<half4*>(b + pl) = l;
            return;
        }
    }
    *reinterpret_cast<uint4*>(y + off) = v.u;
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
    Vec16<T> v;
    v.u = make_uint4(0, 0, 0, 0);
    return v;
}

// erf by Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7) on v_rcp_f32 / v_exp_f32: ~16 VALU
// instructions instead of the ~150 of libm's branchy erff, which made the GELU epilogue of the ConvNeXt
// fc1 GEMMs VALU-bound (profiles/r01a).
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float p = fmaf(1.061405429f, t, -1.453152027f);
    p = fmaf(p, t, 1.421413741f);
    p = fmaf(p, t, -0.284496736f);
    p = fmaf(p, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * ax * ax);
    const float r = 1.0f - p * t * e;
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f)); }

// GELU for fp16 storage: erf(x / sqrt 2) ~ xc * r(xc^2), xc = x clamped to +-4.4, r of degree 8 (minimax LP fit,
// DESIGN.md section 5); max |GELU error| 4.1e-5 in fp32 Horner = 1/12 of an fp16 ulp at |x| ~ 1.
// No transcendental: a wave64 VALU instruction costs 4 cycles per SIMD on gfx950 and the exact-erf form above made the
// fc1 epilogues and the GroupNorm+GELU passes VALU bound.  Written on 2-vectors for historical reasons: the library is
// built with `-target-feature -packed-fp32-ops` (givepose_amd/build.py), so every operation below is a plain
// v_fma_f32 / v_mul_f32 per element -- same roundings, same results.  The fp32 storage path keeps gelu_erf.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Workgroup id -> work item so that each XCD (workgroup ids equal mod 8 share one, round-robin placement) owns ONE contiguous
// run of the n items: XCD x of 8 gets items [x n/8, (x+1) n/8) (bijective for any n).  Every kernel of the ConvNeXt block
// chain (depth-wise conv, fused MLP, fc1, fc2) uses this order over the rows of the activation matrix, so the rows an XCD
// writes are the rows its workgroups read in the next launch (they are still in its private L2).  Speed only.
__device__ __forceinline__ int xcd_chunk(int bid, int n) {
    const int q = n >> 3, r8 = n & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
}

// compile-time loop: f(std::integral_constant<int, I>{}) for I in [I0, N) (bodies that need constexpr indices)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// GELU for fp16 storage: gelu(x) = x * Phi(x), Phi(x) = 0.5 + xc * h(xc^2), xc = clamp(x, +-4.4); 2 h = r, the degree-8
// minimax fit of erf(x / sqrt 2) / x in x^2 (LP fit, weighted by the GELU error x^2 / 2, constrained to meet erf at the
// clamp so that no second clamp is needed: beyond it the result is x * (1 - 5.4e-6) resp. x * 5.4e-6).  The halved
// coefficients are exact (powers of two), so h is bit for bit r / 2.  12 operations per element: clamp, square, 8 FMAs
// (Horner), Phi = fma(xc, h, 0.5), x * Phi  (until round 2: 13 -- 0.5 x and x/2 * erf + x/2 as separate steps).
constexpr float GELU_H[9] = {0.5f * 6.9778819482e-11f, 0.5f * -7.3778779375e-09f, 0.5f * 3.4381198132e-07f, 0.5f * -9.3718131897e-06f,
                             0.5f * 1.6778340171e-04f, 0.5f * -2.1052074914e-03f, 0.5f * 1.9270481587e-02f, 0.5f * -1.3212860816e-01f,
                             0.5f * 7.9751050727e-01f};
__device__ __forceinline__ f32x2 gelu_poly2(f32x2 x) {
    f32x2 xc;
    xc[0] = __builtin_amdgcn_fmed3f(x[0], -4.4f, 4.4f);
    xc[1] = __builtin_amdgcn_fmed3f(x[1], -4.4f, 4.4f);
    const f32x2 t = xc * xc;
    f32x2 p = __builtin_elementwise_fma(f32x2{GELU_H[0], GELU_H[0]}, t, f32x2{GELU_H[1], GELU_H[1]});
#pragma unroll
    for (int k = 2; k < 9; ++k) p = __builtin_elementwise_fma(p, t, f32x2{GELU_H[k], GELU_H[k]});
    return x * __builtin_elementwise_fma(xc, p, f32x2{0.5f, 0.5f});
}
// the same polynomial on NC 2-vectors walked in lock step: NC independent dependency chains, so the FMAs issue back to
// back instead of waiting out each other's latency (fused MLP kernel, csrc/mlp.hip)
template <int NC>
__device__ __forceinline__ void gelu_poly2_xn(f32x2* x) {
    f32x2 xc[NC], t[NC], p[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        xc[i][0] = __builtin_amdgcn_fmed3f(x[i][0], -4.4f, 4.4f);
        xc[i][1] = __builtin_amdgcn_fmed3f(x[i][1], -4.4f, 4.4f);
    }
#pragma unroll
    for (int i = 0; i < NC; ++i) t[i] = xc[i] * xc[i];
#pragma unroll
    for (int i = 0; i < NC; ++i) p[i] = __builtin_elementwise_fma(f32x2{GELU_H[0], GELU_H[0]}, t[i], f32x2{GELU_H[1], GELU_H[1]});
#pragma unroll
    for (int k = 2; k < 9; ++k)
#pragma unroll
        for (int i = 0; i < NC; ++i) p[i] = __builtin_elementwise_fma(p[i], t[i], f32x2{GELU_H[k], GELU_H[k]});
#pragma unroll
    for (int i = 0; i < NC; ++i) x[i] = x[i] * __builtin_elementwise_fma(xc[i], p[i], f32x2{0.5f, 0.5f});
}
__device__ __forceinline__ void gelu_poly2_x8(f32x2 (&x)[8]) { gelu_poly2_xn<8>(x); }
// the same polynomial as 12 separable slices of one operation per element, for callers that hide it in the shadow of
// MFMAs a slice at a time (gemm_wreg_kernel): slice S of chain c works on x[c] with the scratch xc[c], t[c], p[c].
// Plain v_fma_f32 / v_mul_f32 from inline asm (the slices date from the time the library still allowed packed fp32 ops:
// beside MFMAs one v_pk_fma_f32 costs ~22 cycles more than two v_fma_f32, MI355X_MICROARCH.md constants table; the asm
// also pins the order).  Same roundings as gelu_poly2 (IEEE fma / mul either way): bitwise the same results.
__device__ __forceinline__ float vfma(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ float vfma_s(float a, float b, float c) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c)); return d; }
__device__ __forceinline__ float vmul(float a, float b) { float d; asm("v_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
constexpr int GELU_SLICES = 12;
template <int S>
__device__ __forceinline__ void gelu_poly2_slice(f32x2& x, f32x2& xc, f32x2& t, f32x2& p, float c1v) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        if constexpr (S == 0) xc[e] = __builtin_amdgcn_fmed3f(x[e], -4.4f, 4.4f);
        else if constexpr (S == 1) t[e] = vmul(xc[e], xc[e]);
        else if constexpr (S == 2) { float d; asm("v_fma_f32 %0, %1, %2, %3" : "=v"(d) : "s"(GELU_H[0]), "v"(t[e]), "v"(c1v)); p[e] = d; }   // c1v = GELU_H[1] in a VGPR (one SGPR per VALU instruction)
        else if constexpr (S >= 3 && S <= 9) p[e] = vfma_s(p[e], t[e], GELU_H[S - 1 < 2 ? 2 : (S - 1 > 8 ? 8 : S - 1)]);
        else if constexpr (S == 10) { float d; asm("v_fma_f32 %0, %1, %2, 0.5" : "=v"(d) : "v"(xc[e]), "v"(p[e])); p[e] = d; }   // Phi
        else if constexpr (S == 11) x[e] = vmul(x[e], p[e]);
    }
}
// Bilinear x2 blend with pinned roundings (gp_upsample_bilinear2x and gp_groupnorm_upsample2x must agree bit for bit, whatever the
// compiler would contract): h = fma(hx, a, lx * b) along the row, then out = fma(hy, h0, ly * h1) across the two rows.
// src = dst * (in - 1) / (out - 1), rounded once; the empty asm keeps the product from being contracted into the `src - floor`
// that follows (which one kernel might get and the other not).
__device__ __forceinline__ float bilerp_src(float ratio, int dst) {
    float s = ratio * (float)dst;
    asm volatile("" : "+v"(s));
    return s;
}
__device__ __forceinline__ float bilerp_h(float hx, float a, float lx, float b) { return __fmaf_rn(hx, a, __fmul_rn(lx, b)); }
__device__ __forceinline__ float bilerp_v(float hy, float h0, float ly, float h1) { return __fmaf_rn(hy, h0, __fmul_rn(ly, h1)); }
// The vertical step of a whole 16-byte vector.  fp16 storage: the fma and the conversion are ONE instruction (v_fma_mix{lo,hi}_f16:
// fp32 fma, result rounded once to fp16), written out here because hipcc forms it in some instantiations and not in others -- which
// differ in the last bit on near-ties (round 4: the two upsample kernels disagreed in 1 value of 7000 before this).
template <typename T>
__device__ __forceinline__ void bilerp_v_vec(Vec16<T>& o, float hy, const float* h0, float ly, const float* h1) {
    if constexpr (sizeof(T) == 2) {
        unsigned int d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float ta = ly * h1[2 * k], tb = ly * h1[2 * k + 1];
            asm("v_fma_mixlo_f16 %0, %1, %2, %3" : "=v"(d[k]) : "v"(hy), "v"(h0[2 * k]), "v"(ta));
            asm("v_fma_mixhi_f16 %0, %1, %2, %3" : "+v"(d[k]) : "v"(hy), "v"(h0[2 * k + 1]), "v"(tb));
        }
        o.u = uint4{d[0], d[1], d[2], d[3]};
    } else {
#pragma unroll
        for (int e = 0; e < Vec16<T>::N; ++e) o.set(e, bilerp_v(hy, h0[e], ly, h1[e]));
    }
}

// GELU for fp16 storage on PACKED fp16 arithmetic (round 5): gelu(x) = max(x, 0) + R(min(|x|, 4)), R(a) = a (Phi(a) - 1) (even in x,
// -> 0 for large a: every error below is ABSOLUTE, of the size of one fp16 rounding of a value in [0.25, 0.5)), R as a degree-7
// polynomial in u = a / 2 - 1 (minimax fit, |R error| 1.95e-4 in exact arithmetic; coefficients <= 0.33 in this variable: Horner in
// fp16 stays well conditioned -- in the variable a the same polynomial has coefficients up to 58 and loses two digits).  13 operations
// per TWO values (cvt_pk, |x|, min, u, 7 FMAs, max, add) against 2 x 12 + a conversion for gelu_poly2, and the result IS the fp16
// output.  Why it matters: on a SIMD every VALU instruction costs ~4.5 issue cycles that no MFMA of either resident wave hides
// (scripts/probes/mfma_valu_coissue.hip, profiles/r05_fc1_cycle_ablation.txt).  scripts/gelu16_fit.py: fit + simulation (rms error for
// x ~ N(0, 1): 3.0e-4 against 1.4e-4 for the exact GELU rounded to fp16); end to end the fp16 mode's pose errors do not move
// (profiles/r05_gelu16_end_to_end.txt).  v_pk_fma_f16 is not the packed-fp32 family of DESIGN.md 6b (profiles/r05_repro_pkh.txt).
// GP_GELU16=0 keeps the fp32 polynomial (A/B switch, read by the launchers).
constexpr float GELU16_C[8] = {-4.5349121094e-02f, 1.7163085938e-01f, -2.2192382812e-01f, -1.4266967773e-02f,
                               3.2568359375e-01f, -2.3901367188e-01f, -5.8441162109e-02f, 8.1665039062e-02f};   // c0 .. c7, fp16 values
__device__ __forceinline__ half2v h2(float v) { return half2v{(half_t)v, (half_t)v}; }
constexpr int GELU16_SLICES = 13;
template <int S>
__device__ __forceinline__ void gelu16_slice(const f32x2& x, half2v& xh, half2v& u, half2v& p) {
    if constexpr (S == 0) xh = half2v{(half_t)x[0], (half_t)x[1]};
    else if constexpr (S == 1) u = __builtin_elementwise_max(xh, -xh);
    else if constexpr (S == 2) u = __builtin_elementwise_min(u, h2(4.0f));
    else if constexpr (S == 3) u = __builtin_elementwise_fma(u, h2(0.5f), h2(-1.0f));
    else if constexpr (S == 4) p = __builtin_elementwise_fma(h2(GELU16_C[7]), u, h2(GELU16_C[6]));
    else if constexpr (S >= 5 && S <= 10) p = __builtin_elementwise_fma(p, u, h2(GELU16_C[10 - S]));
    else if constexpr (S == 11) xh = __builtin_elementwise_max(xh, h2(0.0f));
    else if constexpr (S == 12) p = xh + p;
}
// NC value pairs walked in lock step (NC independent chains); out[i] = the two fp16 results of x[i], packed
template <int NC>
__device__ __forceinline__ void gelu16_xn(const f32x2* x, unsigned* out) {
    half2v xh[NC], u[NC], p[NC];
    static_for<0, GELU16_SLICES>([&](auto sc) {
#pragma unroll
        for (int i = 0; i < NC; ++i) gelu16_slice<decltype(sc)::value>(x[i], xh[i], u[i], p[i]);
    });
#pragma unroll
    for (int i = 0; i < NC; ++i) out[i] = __builtin_bit_cast(unsigned, p[i]);
}
static inline bool gp_gelu16_enabled() {   // host side
    static const bool on = [] { const char* e = getenv("GP_GELU16"); return !(e && e[0] == '0'); }();
    return on;
}

__device__ __forceinline__ float gelu_poly1(float x) { return gelu_poly2(f32x2{x, x})[0]; }

template <typename T> __device__ __forceinline__ float gelu_for(float v) {
    if constexpr (sizeof(T) == 2) return gelu_poly1(v); else return gelu_erf(v);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case GP_ACT_GELU: return gelu_erf(v);
        case GP_ACT_RELU: return fmaxf(v, 0.0f);
        case GP_ACT_LRELU: return v > 0.0f ? v : 0.1f * v;
        default: return v;
    }
}

// Butterfly reductions over the `width` (power of two, <= 64) consecutive lanes that contain this lane.  Inside a
// row of 16 lanes the exchange is a DPP operand modifier of a VALU op (quad_perm xor 1 / xor 2, row_half_mirror,
// row_mirror: after the two quad steps every lane holds its quad's total, so mirroring adds the partner quad / octet);
// only the 16- and 32-lane steps go through the LDS crossbar (ds_bpermute via __shfl_xor).
#define GP_DPP_STEP(op, ctrl) v = op(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, false)))
__device__ __forceinline__ float gp_addf(float a, float b) { return a + b; }
__device__ __forceinline__ float group_sum(float v, int width) {
    if (width >= 2) GP_DPP_STEP(gp_addf, 0xB1);    // quad_perm [1,0,3,2]
    if (width >= 4) GP_DPP_STEP(gp_addf, 0x4E);    // quad_perm [2,3,0,1]
    if (width >= 8) GP_DPP_STEP(gp_addf, 0x141);   // row_half_mirror
    if (width >= 16) GP_DPP_STEP(gp_addf, 0x140);  // row_mirror
    if (width >= 32) v += __shfl_xor(v, 16, 64);
    if (width >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float group_max(float v, int width) {
    if (width >= 2) GP_DPP_STEP(fmaxf, 0xB1);
    if (width >= 4) GP_DPP_STEP(fmaxf, 0x4E);
    if (width >= 8) GP_DPP_STEP(fmaxf, 0x141);
    if (width >= 16) GP_DPP_STEP(fmaxf, 0x140);
    if (width >= 32) v = fmaxf(v, __shfl_xor(v, 16, 64));
    if (width >= 64) v = fmaxf(v, __shfl_xor(v, 32, 64));
    return v;
}

// 256 zero bytes per translation unit: source of LDS-DMA lanes that fall in padding / out of range
static __device__ __attribute__((aligned(256), used)) unsigned int gp_zero_page_tu[64];

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
