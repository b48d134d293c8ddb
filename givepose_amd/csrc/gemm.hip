// MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] )
//
// All kernels below stage both operands through LDS with LDS-DMA (global_load_lds_dwordx4) as [row][RB bytes] with a
// 16-byte-chunk XOR swizzle applied on the SOURCE address, so that the ds_read_b128 fragment reads are bank-conflict
// free.  W rows are the MFMA "A" operand and X rows the "B" operand, so a lane's four accumulator registers are four
// consecutive n of one m.  In conv mode the X loader walks (kh,kw,ci) in K order and takes padding from a zero page,
// so no im2col buffer ever exists in HBM.  Workgroup ids are remapped so that each XCD (ids equal mod 8 share one)
// owns a contiguous run of tiles with n fastest: the X panel of an m-tile is fetched into one XCD's L2 once.
// Split-K (skinny GEMMs: M = 64 rows of the PnP fc layers) runs on the 128x128 LDS-DMA tile with fp32 partial slabs
// and a reduce kernel that applies the epilogue.  The round-1 register-staged 128x128 kernel is gone: its
// compiler-scheduled ds_read_b128 + MFMA loop corrupted packed-fp32 VALU results of OTHER kernels' waves resident on
// the same SIMD (two-stream reproducer scripts/race_min.py).
//
// dtype f16: v_mfma_f32_16x16x32_f16 (8 halfs = one 16-B chunk per lane per MFMA).
// dtype f32: v_mfma_f32_16x16x4_f32 x4 per 16-B chunk (exact fp32 products, fp32 accumulate); the
//            k order inside a chunk is permuted identically for both operands, which is harmless.
#include "common.hpp"

namespace {

struct GemmKP {
    const void* X;
    const void* W;
    const float* bias;
    const float* gamma;
    const void* res;
    void* C;
    float* ws;
    int M, N, K, ldx, ldc, ldres, epi, out_f32;
    int conv, H, Win, Cin, KW, stride, pad, Ho, Wo;
    int tiles_m, tiles_n, nkt, kt_per_split, splitk;
    float* gn_partial;  // fused GroupNorm statistics (large-tile kernels): (B, HW/64, G, 2) chunk sums, or null
    int gn_cpg, gn_hw;
    int early;               // gemm_wreg3_kernel: start the first tile while the weight slice streams in (round 6)
    const float* ln_stats;   // GP_EPI_LNFOLD_GELU: (M, 2, ln_nsl) partial row moments, ln_s (N) column sums of W
    const float* ln_s;
    int ln_nsl;
    float ln_eps;
    int dbg;  // timing-only ablations of the large-tile kernel: 1 = no in-loop DMA, 2 = no MFMA/LDS reads (wrong results)
    const char* pf;   // gp_gemm_desc.prefetch: bytes [0, pf_bytes) touched by the workgroups as they start (hint)
    long pf_bytes;
    // split-operand mode (gp_gemm_desc.split_shift > 0): X and W are fp16 PLANES, hi at the pointer and lo' = fp16((v - hi) * 2^S)
    // xplane_b / wplane_b bytes behind it; the K loop runs three segments of split_n1 steps each -- X_hi W_lo', X_lo' W_hi,
    // then (accumulators scaled by split_scale = 2^-S) X_hi W_hi -- so every fp16 schedule of this file computes
    // x w = x_hi w_hi + 2^-S (x_hi w_lo' + x_lo' w_hi) to ~2^-22 relative with fp32 accumulation.  Output and residual are fp32.
    int split_n1;
    long xplane_b, wplane_b;
    float split_scale;
    // gp_gemm_desc.out_planes: C is written as the hi / lo' planes the NEXT split-operand GEMM reads (lo' cplane elements behind hi)
    int out_planes;
    long cplane;
    float split_up;   // 2^S
    // gp_gemm_desc.c16: besides the fp32 output C, the same values rounded to fp16 at c16 (row stride ldc16) -- the fp32 residual
    // stream of the fp16 mode: the stream is accumulated in fp32, the next block's depth-wise conv reads the rounded copy
    void* c16;
    int ldc16;
};

// Investigation build only (GP_EXTRA_HIPCC_FLAGS=-DGP_CLOCK_STAMPS GP_BUILD_TAG=clk, scripts/kernel_clock.py): thread 0 of every workgroup
// stores (s_memtime, s_memrealtime) around the kernel's main loop into the workspace, which nothing else reads in these launches
// (no split-K): in-kernel shader clock = d memtime / d memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6).
__device__ __forceinline__ void clk_stamp(const GemmKP& p, int which) {
#ifdef GP_CLOCK_STAMPS
    if (threadIdx.x == 0 && p.ws && p.splitk <= 1) {
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = reinterpret_cast<unsigned long long*>(p.ws) + ((long)blockIdx.x * 2 + which) * 2;
        o[0] = t; o[1] = r;
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
}

template <typename T>
__device__ __forceinline__ void epi_store(const GemmKP& p, int m, int n, f32x4 v) {
    if (p.bias) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + n);
        v += b;
    }
    switch (p.epi) {
        case GP_EPI_GELU:
            for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            break;
        case GP_EPI_RELU:
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
            break;
        case GP_EPI_LRELU:
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.1f * v[j];
            break;
        case GP_EPI_RES_RELU: {
            const T* r = reinterpret_cast<const T*>(p.res) + (long)m * p.ldres + n;
            for (int j = 0; j < 4; ++j) v[j] = fmaxf((float)r[j] + v[j], 0.0f);
            break;
        }
        case GP_EPI_SCALE_RES: {
            const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + n);
            const T* r = reinterpret_cast<const T*>(p.res) + (long)m * p.ldres + n;
            if constexpr (sizeof(T) == 2) {
                const half4 rv = *reinterpret_cast<const half4*>(r);
                for (int j = 0; j < 4; ++j) v[j] = (float)rv[j] + g[j] * v[j];
            } else {
                const f32x4 rv = *reinterpret_cast<const f32x4*>(r);
                for (int j = 0; j < 4; ++j) v[j] = rv[j] + g[j] * v[j];
            }
            break;
        }
        default: break;
    }
    if (p.out_f32 || sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = v;
    } else {
        half4 o;
        for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
        *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.C) + (long)m * p.ldc + n) = o;
    }
}

// ---- epilogue pieces with every global LOAD hoisted ahead of the first global STORE of a tile.
// gfx950's vmcnt counts loads and stores together, in issue order: a load whose data is waited for after a
// store also waits for that store's acknowledgement (~1 us under load).  The first version loaded bias /
// gamma / residual per 4-element group between the stores and spent ~15 us per 256x256 tile doing so
// (scripts/gemm_bench.py abl.*, K = 64).
template <typename T> struct Res4 { typedef half4 type; };
template <> struct Res4<float> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ typename Res4<T>::type load_res4(const GemmKP& p, int m, int n) {
    return *reinterpret_cast<const typename Res4<T>::type*>(reinterpret_cast<const T*>(p.res) + (long)m * p.ldres + n);
}

template <typename T>
__device__ __forceinline__ f32x4 epi_apply(int epi, f32x4 v, const f32x4& b, const f32x4& g, const typename Res4<T>::type& r) {
    v += b;
    switch (epi) {
        case GP_EPI_GELU:
            if constexpr (sizeof(T) == 2) {
                const f32x2 lo = gelu_poly2(f32x2{v[0], v[1]}), hi = gelu_poly2(f32x2{v[2], v[3]});
                v = f32x4{lo[0], lo[1], hi[0], hi[1]};
            } else {
                for (int j = 0; j < 4; ++j) v[j] = gelu_erf(v[j]);
            }
            break;
        case GP_EPI_RELU:
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
            break;
        case GP_EPI_LRELU:
            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.1f * v[j];
            break;
        case GP_EPI_SCALE_RES:
            for (int j = 0; j < 4; ++j) v[j] = (float)r[j] + g[j] * v[j];
            break;
        case GP_EPI_RES_RELU:
            for (int j = 0; j < 4; ++j) v[j] = fmaxf((float)r[j] + v[j], 0.0f);
            break;
        default: break;
    }
    return v;
}

template <typename T> __device__ __forceinline__ void store4(const GemmKP& p, int m, int n, const f32x4& v) {
    if (p.out_f32 || sizeof(T) == 4) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n) = v;
    } else {
        half4 o;
        for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
        *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.C) + (long)m * p.ldc + n) = o;
    }
}

template <typename T> __device__ __forceinline__ void mma(f32x4& acc, const uint4& a, const uint4& b);
template <> __device__ __forceinline__ void mma<half_t>(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&a),
                                                 *reinterpret_cast<const half8*>(&b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<float>(f32x4& acc, const uint4& a, const uint4& b) {
    const float* af = reinterpret_cast<const float*>(&a);
    const float* bf = reinterpret_cast<const float*>(&b);
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc, 0, 0, 0);
}

template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmKP p) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int n4 = p.N >> 2;
    if (i >= (long)p.M * n4) return;
    const int m = (int)(i / n4), n = (int)(i - (long)m * n4) * 4;
    f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < p.splitk; ++s) v += *reinterpret_cast<const f32x4*>(p.ws + ((long)s * p.M + m) * p.N + n);
    epi_store<T>(p, m, n, v);
}


// =====================================================================================================
// Large-tile variant: BM = WM*MT*16 pixels x BN = WN*NT*16 channels, 8 waves, K step 128 bytes.
//   A: 256 x 256 (waves 2m x 4n, wave tile 128m x 64n)      B: 256 x 128 (waves 4m x 2n, wave tile 64 x 64)
// Per FLOP it moves half (A) / three quarters (B) of the 128x128 kernel's L2->LDS bytes and, with the
// 128x64 wave tile, 25 % fewer LDS fragment bytes -- the 128x128 kernel is LDS-pipe bound (ds_write of the
// register-staged tiles + fragment reads exceed the MFMA time, profiles/r01a).  Global->LDS goes through
// LDS-DMA (global_load_lds_dwordx4): no staging VGPRs, no ds_write; the LDS image stays lane-linear and the
// bank-conflict swizzle is applied to the per-lane SOURCE address (chunk ^= row & 7), the same involution as
// on the fragment reads.  Padding / out-of-range rows read a zero page.  Two LDS stages: the DMA of step
// t+1 is issued before the MFMAs of step t and drained (vmcnt(0)) right before the barrier that ends step t.
// Epilogue: each wave transposes its accumulators through a private 8 KB LDS slab (32 rows x 64 fp32,
// XOR-swizzled) so that global stores/residual loads are whole 128-B row segments.

// ---- lean epilogue of the large-tile kernels: fp16 output, tile fully inside C.
// The activation runs in the MFMA register layout (lane = 4 consecutive n of one m, so bias / gamma are 4 registers
// per n-tile), the result is packed to fp16 BEFORE the transpose (half the LDS bytes of the fp32 slab), rows are
// read back 16 B per lane and stored 8 rows x 128 B per wave instruction; the residual (fp16, prefetched ahead of
// the first store) is added with packed fp16 adds.  EPI is a template parameter: the run-time switch of the generic
// epilogue below cost more (branches around every 4-element group) than its arithmetic.
template <int MT, int NT, int EPI>
__device__ __forceinline__ void epilogue_lean(const GemmKP& p, f32x4 (&acc)[NT][MT], char* slab, int mb, int nb, int lane) {
    constexpr int PITCH = 144, NH = MT / 4;   // 64 rows x (128 B + 16 B pad) per wave and half tile
    constexpr bool RES = EPI == GP_EPI_SCALE_RES || EPI == GP_EPI_RES_RELU;
    const int fr = lane & 15, fq = lane >> 4, rr = lane >> 3, rc = lane & 7;
    f32x4 g4[NT];   // (the bias is already in the accumulators)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nb + nt * 16 + fq * 4;
        g4[nt] = EPI == GP_EPI_SCALE_RES ? *reinterpret_cast<const f32x4*>(p.gamma + n) : f32x4{1.f, 1.f, 1.f, 1.f};
    }
    half8 rres[RES ? NH : 1][8];
    if constexpr (RES) {
        const half_t* R = reinterpret_cast<const half_t*>(p.res);
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                rres[h][i] = *reinterpret_cast<const half8*>(R + (long)(mb + h * 64 + i * 8 + rr) * p.ldres + nb + rc * 8);
    }
    // LayerNorm folded into the epilogue: v = rstd[m] * acc + (bias[n] - rstd[m] * mean[m] * colsum[n])
    constexpr bool LNF = EPI == GP_EPI_LNFOLD_GELU;
    f32x4 lc4[LNF ? NT : 1], ls4[LNF ? NT : 1];
    float lr[LNF ? MT : 1], lmr[LNF ? MT : 1];
    if constexpr (LNF) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = nb + nt * 16 + fq * 4;
            lc4[nt] = *reinterpret_cast<const f32x4*>(p.bias + n);
            ls4[nt] = *reinterpret_cast<const f32x4*>(p.ln_s + n);
        }
        const float invK = 1.0f / p.K;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float* st = p.ln_stats + (long)(mb + mt * 16 + fr) * 2 * p.ln_nsl;
            float su = 0.f, sq = 0.f;
            if (p.ln_nsl == 4) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(st), b = *reinterpret_cast<const f32x4*>(st + 4);
                su = (a[0] + a[1]) + (a[2] + a[3]);
                sq = (b[0] + b[1]) + (b[2] + b[3]);
            } else {
                for (int i = 0; i < p.ln_nsl; ++i) { su += st[i]; sq += st[p.ln_nsl + i]; }
            }
            const float mu = su * invK;
            lr[mt] = __builtin_amdgcn_rsqf(fmaxf(sq * invK - mu * mu, 0.f) + p.ln_eps);
            lmr[mt] = -mu * lr[mt];
        }
    }
    half_t* C = reinterpret_cast<half_t*>(p.C);
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        float gs[NT], gq[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) gs[nt] = gq[nt] = 0.f;
#pragma unroll
        for (int ml = 0; ml < 4; ++ml)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                f32x4 v = acc[nt][h * 4 + ml];
                if constexpr (LNF) {
                    const float r = lr[h * 4 + ml], mr = lmr[h * 4 + ml];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaf(r, v[j], fmaf(mr, ls4[nt][j], lc4[nt][j]));
                }
                if constexpr (EPI == GP_EPI_GELU || LNF) {
                    const f32x2 lo = gelu_poly2(f32x2{v[0], v[1]}), hi = gelu_poly2(f32x2{v[2], v[3]});
                    v = f32x4{lo[0], lo[1], hi[0], hi[1]};
                } else if constexpr (EPI == GP_EPI_RELU) {
                    for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.0f);
                } else if constexpr (EPI == GP_EPI_LRELU) {
                    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.0f ? v[j] : 0.1f * v[j];
                } else if constexpr (EPI == GP_EPI_SCALE_RES) {
                    v *= g4[nt];
                }
                if (!RES && p.gn_partial) {
                    gs[nt] += (v[0] + v[1]) + (v[2] + v[3]);
                    gq[nt] += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
                }
                half4 o;
                for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
                *reinterpret_cast<half4*>(slab + (ml * 16 + fr) * PITCH + (nt * 4 + fq) * 8) = o;
            }
        if (!RES && p.gn_partial) {   // fused GroupNorm statistics: (sum, sum of squares) per 64 rows and channel group
            const int mrow = mb + h * 64;
            const int G = p.N / p.gn_cpg, cpi = p.gn_hw >> 6;
            const int b = mrow / p.gn_hw, ch = (mrow - b * p.gn_hw) >> 6;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                float a = gs[nt], q = gq[nt];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
                if (p.gn_cpg == 8) { a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64); }
                if (fr == 0 && (p.gn_cpg == 4 || (fq & 1) == 0)) {
                    float* o = p.gn_partial + (((long)b * cpi + ch) * G + (nb + nt * 16 + fq * 4) / p.gn_cpg) * 2;
                    o[0] = a;
                    o[1] = q;
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = i * 8 + rr;
            half8 v = *reinterpret_cast<const half8*>(slab + row * PITCH + rc * 16);
            if constexpr (RES) v += rres[h][i];
            if constexpr (EPI == GP_EPI_RES_RELU) v = __builtin_elementwise_max(v, half8{0, 0, 0, 0, 0, 0, 0, 0});
            if (p.dbg != 4 || (float)v[0] == 12345.0f)
                *reinterpret_cast<half8*>(C + (long)(mb + h * 64 + row) * p.ldc + nb + rc * 8) = v;
        }
    }
}

typedef __attribute__((address_space(3))) char lds_char_t;

// gp_gemm_desc.prefetch: the launch's workgroups touch [pf, pf + pf_bytes) once, 1 KB per wave instruction (16 B per lane),
// at most NPF instructions per wave.  The loads have register destinations that nothing computes with; prefetch_issue() returns
// them and prefetch_retire() "uses" them behind the kernel's first vmcnt wait (vmcnt retires in issue order and these loads are
// older than everything the kernel waits for, so the use costs no stall).  They are ORDINARY loads the compiler knows about:
// until round 4 they were inline asm, invisible to hipcc's wait-count tracking -- which left it free to move the (as it
// believed, already written) destination registers' value elsewhere and reuse the registers while the load was still in
// flight.  In the small-M kernel it did: the landing prefetch overwrote the K-loop counter (memory fault on the one launch of the
// step whose prefetch is long enough for a second load per wave, found by scripts/b1_eager_probe.py).
constexpr int NPF = 4;
typedef unsigned pf_u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: no 'v' constraint)
struct PfSink { pf_u32x4 r0, r1, r2, r3; };
__device__ __forceinline__ void prefetch_one(const GemmKP& p, long i, long hi, int lane, pf_u32x4& r) {
    const long off = (i << 10) + lane * 16;
    r = pf_u32x4{0u, 0u, 0u, 0u};
    if (i < hi && off + 16 <= p.pf_bytes) r = *reinterpret_cast<const pf_u32x4*>(p.pf + off);
}
__device__ __forceinline__ PfSink prefetch_issue(const GemmKP& p, int bid, int nwg, int wave, int nwaves, int lane) {
    PfSink s;
    s.r0 = s.r1 = s.r2 = s.r3 = pf_u32x4{0u, 0u, 0u, 0u};
    if (p.pf) {
        const long pieces = (p.pf_bytes + 1023) >> 10, per = (pieces + nwg - 1) / nwg;
        const long lo = (long)bid * per, hi = min(pieces, lo + per);
        prefetch_one(p, lo + wave, hi, lane, s.r0);
        prefetch_one(p, lo + wave + nwaves, hi, lane, s.r1);
        prefetch_one(p, lo + wave + 2 * nwaves, hi, lane, s.r2);
        prefetch_one(p, lo + wave + 3 * nwaves, hi, lane, s.r3);
    }
    return s;
}
__device__ __forceinline__ void prefetch_retire(const PfSink s) {
    asm volatile("" ::"v"(s.r0), "v"(s.r1), "v"(s.r2), "v"(s.r3));
}

// One LDS-DMA wave instruction: 64 lanes x 16 B from per-lane global addresses to LDS [lds_addr, +1 KB).
// Issued from inline asm so that hipcc does not count it: with the builtin form hipcc puts an
// s_waitcnt vmcnt(0) in front of the first ds_read after it (it cannot tell the DMA's LDS destination from the
// buffer being read), which serialises the DMA of step t+1 with the MFMAs of step t.  The kernel waits for
// these loads itself (s_waitcnt vmcnt(0) before the barrier that publishes the stage).  M0 carries the LDS
// destination and is restored (hipcc reserves it).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

// LDS-DMA with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset: a K step then costs one
// scalar add per operand instead of a 64-bit vector add + select per instruction (the load phase of the ping-pong
// schedule issues its VALU work beside the partner wave's MFMAs, at half rate)
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

// ---- generic epilogue of the tile kernels (any output type, edge tiles, split-operand mode), through a wave-private LDS
//      slab (8 KB).  RT = element type of residual and output (fp32 in the split-operand mode); mb / nb = first row / column
//      of the WAVE's tile.
// EPI_CT >= 0: the epilogue code is known at compile time (the split-operand kernels dispatch on it once per workgroup: the
// run-time switch around every 4-element group costs more than the arithmetic, section 5.1 of DESIGN.md); -1: read p.epi.
template <typename RT, int MT, int NT, bool SPL, int EPI_CT = -1>
__device__ __forceinline__ void epilogue_generic(const GemmKP& pin, f32x4 (&acc)[NT][MT], char* slab, int mb, int nb, int lane) {
    const GemmKP& p = pin;
    const int epi_code = EPI_CT >= 0 ? EPI_CT : pin.epi;
    // ---- epilogue through a wave-private LDS slab: 32 rows (m) x 64 fp32 (n), 16-B chunk ^= row & 7.
    //      Lane owns output columns n .. n+3 (fixed) of rows i*4 + (lane>>4); every global load (bias, gamma,
    //      the whole residual tile in f16 mode) is issued before the first store.
    const int fr = lane & 15, fq = lane >> 4;
    const int en = nb + (lane & 15) * 4, er = lane >> 4;
    const bool nok = en < p.N;
    const f32x4 zero4 = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 b4 = (p.bias && nok) ? *reinterpret_cast<const f32x4*>(p.bias + en) : zero4;
    const bool sres = epi_code == GP_EPI_SCALE_RES || epi_code == GP_EPI_RES_RELU;
    const f32x4 g4 = (epi_code == GP_EPI_SCALE_RES && nok) ? *reinterpret_cast<const f32x4*>(p.gamma + en) : zero4;
    constexpr bool PRE = sizeof(RT) == 2;           // f16: prefetch the whole residual tile (64 VGPRs at MT = 8)
    typename Res4<RT>::type r4[PRE ? MT / 2 : 1][8];
    if (PRE) {
#pragma unroll
        for (int j = 0; j < MT / 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int m = mb + j * 32 + i * 4 + er;
                for (int e = 0; e < 4; ++e) r4[PRE ? j : 0][i][e] = 0;
                if (sres && nok && m < p.M) r4[PRE ? j : 0][i] = load_res4<RT>(p, m, en);
            }
    }
    float gsum = 0.f, gsq = 0.f;
#pragma unroll
    for (int j = 0; j < MT / 2; ++j) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int row = h * 16 + fr;
                *reinterpret_cast<f32x4*>(slab + row * 256 + (((nt * 4 + fq) ^ (row & 7)) << 4)) = acc[nt][2 * j + h];
            }
#pragma unroll
        for (int i0 = 0; i0 < 8; i0 += 4) {
            if (!PRE) {   // fp32 storage: residual in batches of 4 rows (register budget)
#pragma unroll
                for (int i = i0; i < i0 + 4; ++i) {
                    const int m = mb + j * 32 + i * 4 + er;
                    for (int e = 0; e < 4; ++e) r4[0][i][e] = 0;
                    if (sres && nok && m < p.M) r4[0][i] = load_res4<RT>(p, m, en);
                }
            }
#pragma unroll
            for (int i = i0; i < i0 + 4; ++i) {
                const int row = i * 4 + er, chunk = lane & 15;
                const f32x4 v = *reinterpret_cast<const f32x4*>(slab + row * 256 + ((chunk ^ (row & 7)) << 4));
                const int m = mb + j * 32 + row;
                if (m < p.M && nok) {
                    const f32x4 o = epi_apply<RT>(epi_code, v, b4, g4, r4[PRE ? j : 0][i]);
                    if (SPL && p.out_planes) {   // hi = fp16(o), lo' = fp16((o - hi) 2^S): what gp_split_planes would make of the fp32 result
                        half4 hv, lv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hv[e] = (half_t)o[e];
                            lv[e] = (half_t)((o[e] - (float)hv[e]) * p.split_up);
                        }
                        half_t* ch = reinterpret_cast<half_t*>(p.C) + (long)m * p.ldc + en;
                        *reinterpret_cast<half4*>(ch) = hv;
                        *reinterpret_cast<half4*>(ch + p.cplane) = lv;
                    } else
                    if (p.dbg != 4 || o[0] == 12345.678f) store4<RT>(p, m, en, o);
                    if (p.c16) {
                        half4 hv;
#pragma unroll
                        for (int e = 0; e < 4; ++e) hv[e] = (half_t)o[e];
                        *reinterpret_cast<half4*>(reinterpret_cast<half_t*>(p.c16) + (long)m * p.ldc16 + en) = hv;
                    }
                    gsum += (o[0] + o[1]) + (o[2] + o[3]);
                    gsq += (o[0] * o[0] + o[1] * o[1]) + (o[2] * o[2] + o[3] * o[3]);
                }
            }
        }
        // fused GroupNorm statistics: one (sum, sum of squares) per 64 output rows and channel group, fixed order
        if (p.gn_partial && (j & 1)) {
            float a = gsum, q = gsq;
            a += __shfl_xor(a, 16, 64); q += __shfl_xor(q, 16, 64);
            a += __shfl_xor(a, 32, 64); q += __shfl_xor(q, 32, 64);
            if (p.gn_cpg == 8) { a += __shfl_xor(a, 1, 64); q += __shfl_xor(q, 1, 64); }
            const int mrow = mb + (j - 1) * 32;
            if (er == 0 && nok && mrow < p.M && (p.gn_cpg == 4 || (lane & 1) == 0)) {
                const int G = p.N / p.gn_cpg, cpi = p.gn_hw >> 6;
                const int b = mrow / p.gn_hw, ch = (mrow - b * p.gn_hw) >> 6;
                float* o = p.gn_partial + (((long)b * cpi + ch) * G + en / p.gn_cpg) * 2;
                o[0] = a;
                o[1] = q;
            }
            gsum = gsq = 0.f;
        }
    }
}

template <int MT, int NT>
__device__ __forceinline__ void epilogue_split(const GemmKP& p, f32x4 (&acc)[NT][MT], char* slab, int mb, int nb, int lane) {
    switch (p.epi) {       // wave-uniform, once per workgroup
        case GP_EPI_GELU: epilogue_generic<float, MT, NT, true, GP_EPI_GELU>(p, acc, slab, mb, nb, lane); break;
        case GP_EPI_RELU: epilogue_generic<float, MT, NT, true, GP_EPI_RELU>(p, acc, slab, mb, nb, lane); break;
        case GP_EPI_LRELU: epilogue_generic<float, MT, NT, true, GP_EPI_LRELU>(p, acc, slab, mb, nb, lane); break;
        case GP_EPI_SCALE_RES: epilogue_generic<float, MT, NT, true, GP_EPI_SCALE_RES>(p, acc, slab, mb, nb, lane); break;
        case GP_EPI_RES_RELU: epilogue_generic<float, MT, NT, true, GP_EPI_RES_RELU>(p, acc, slab, mb, nb, lane); break;
        default: epilogue_generic<float, MT, NT, true, GP_EPI_NONE>(p, acc, slab, mb, nb, lane); break;
    }
}

template <typename T, int WM, int WN, int MT, int NT, int NS, bool DB = false, int RB = 128, bool PP = false, bool SPL = false, bool R32 = false>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 4 ? 2 : 1) void gemm_big_kernel(const GemmKP pin) {
    static_assert(NT == 4, "epilogue slab assumes a 64-wide wave tile");
    static_assert(!SPL || sizeof(T) == 2, "split-operand mode: fp16 planes");
    static_assert(!R32 || (sizeof(T) == 2 && !SPL), "R32: fp16 operands with an fp32 residual / output");
    typedef typename std::conditional<SPL || R32, float, T>::type RT;   // residual / output element (SPL, R32: fp32)
    // split-K (generic ring schedule only): workgroup row blockIdx.y multiplies K steps [kt0, kt0 + nkt) into its own
    // fp32 slab of the workspace; the host passes C = workspace, out_f32, no bias / epilogue (splitk_reduce_kernel applies them)
    GemmKP p = pin;
    int kt0 = 0;
    if constexpr (!PP && !(DB && NS == 2)) {
        if (pin.splitk > 1) {
            kt0 = blockIdx.y * pin.kt_per_split;
            p.nkt = min(pin.nkt - kt0, pin.kt_per_split);
            p.C = reinterpret_cast<float*>(pin.C) + (long)blockIdx.y * pin.M * pin.N;
        }
    }
    static_assert(!PP || (WM * WN == 8 && (NS == 4 || NS == 5) && RB == 64 && !DB), "ping-pong schedule: 8 waves, 4/5-stage ring of 64-byte K steps");
    constexpr int NW = WM * WN;
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    // RB = bytes of K per LDS row and per K step: 128 (two MFMA k-groups per step) or 64 (one; half the stage
    // size, so twice the stages / DMA bytes in flight fit the 160 KB LDS)
    static_assert(RB == 128 || (RB == 64 && !DB), "RB");
    constexpr int EPT = 16 / sizeof(T), KPT = RB / sizeof(T), CPR = RB / 16, RPI = 1024 / RB;
    constexpr int XI = BM / RPI / NW, WI = BN / RPI / NW;  // LDS-DMA instructions per wave per K step
    constexpr int STAGE = (BM + BN) * RB;
    static_assert(NS * STAGE >= NW * 8192, "epilogue slabs must fit");
    __shared__ __attribute__((aligned(1024))) char smem[NS * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wn = wave / WM;

    const int nblk = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const PfSink pfs = prefetch_issue(p, bid + blockIdx.y * gridDim.x, gridDim.x * gridDim.y, wave, WM * WN, lane);   // gp_gemm_desc.prefetch (hint)

    const T* __restrict__ X = reinterpret_cast<const T*>(p.X);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W);
    const T* zero = reinterpret_cast<const T*>(gp_zero_page_tu);

    // ---- DMA source state: instruction i of this wave fills rows (i*NW + wave)*8 .. +8 of a tile;
    //      lane -> row +(lane>>3), LDS chunk lane&7 holds logical chunk (lane&7) ^ (row&7).
    //      Per lane: one base pointer per row (tap (0,0) / k = 0) and a bit mask of the in-bounds filter taps, so a
    //      K step costs a scalar offset + add + select per DMA (the first version recomputed (hi, wi) and a 64-bit
    //      address per DMA: 1.8 VALU ops per MFMA, which competes with the MFMAs for the SIMD's issue slots).
    const int lrow = lane / CPR;
    const int lchunk = RB == 128 ? ((lane & 7) ^ (lrow & 7)) : ((lane & 3) ^ ((-(lrow >> 2)) & 3));
    const char* xptr[XI];
    unsigned xmask[XI];
    const char* wptr[WI];
#pragma unroll
    for (int i = 0; i < XI; ++i) {
        const int m = m0 + (i * NW + wave) * RPI + lrow;
        const bool ok = m < p.M;
        if (p.conv) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, r = m - b * hw;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            xptr[i] = reinterpret_cast<const char*>(X + (((long)b * p.H + hi0) * p.Win + wi0) * p.Cin + lchunk * EPT);
            unsigned mk = 0;
            const int KH = p.K / (p.KW * p.Cin);
            for (int kh = 0; kh < KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw)
                    if (ok && (unsigned)(hi0 + kh) < (unsigned)p.H && (unsigned)(wi0 + kw) < (unsigned)p.Win)
                        mk |= 1u << (kh * p.KW + kw);
            xmask[i] = mk;
        } else {
            xptr[i] = reinterpret_cast<const char*>(X + (long)m * p.ldx + lchunk * EPT);
            xmask[i] = ok ? 1u : 0u;
        }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int n = n0 + (i * NW + wave) * RPI + lrow;
        wptr[i] = n < p.N ? reinterpret_cast<const char*>(W + (long)n * p.K + lchunk * EPT) : nullptr;
    }
    const int cpt = p.conv ? p.Cin / KPT : 1;

    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;
    auto stage = [&](int buf, int kt) {
        const unsigned xs = lds0 + buf * STAGE + wave * 1024;
        const unsigned ws = xs + BM * RB;
        long xoff;      // wave-uniform byte offset of this K step from the row base pointer
        unsigned bit = 1u;
        kt += kt0;
        long xadd = 0, wadd = 0;
        if constexpr (SPL) {   // segment of the concatenated K: 0 = X_hi W_lo', 1 = X_lo' W_hi, 2 = X_hi W_hi
            const int seg = (kt >= p.split_n1 ? 1 : 0) + (kt >= 2 * p.split_n1 ? 1 : 0);
            kt -= seg * p.split_n1;
            xadd = seg == 1 ? p.xplane_b : 0;
            wadd = seg == 0 ? p.wplane_b : 0;
        }
        if (p.conv) {
            const int tap = kt / cpt;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            xoff = (((long)kh * p.Win + kw) * p.Cin + (kt - tap * cpt) * KPT) * (long)sizeof(T);
            bit = 1u << tap;
        } else {
            xoff = (long)kt * RB;
        }
        const char* zp = reinterpret_cast<const char*>(zero);
#pragma unroll
        for (int i = 0; i < XI; ++i) glds16((xmask[i] & bit) ? xptr[i] + xoff + xadd : zp, xs + i * NW * 1024);
#pragma unroll
        for (int i = 0; i < WI; ++i) glds16(wptr[i] ? wptr[i] + (long)kt * RB + wadd : zp, ws + i * NW * 1024);
    };

    // the lean epilogue (fp16 output, tile fully inside C: see epilogue_lean) takes the bias as the initial value of
    // the accumulators: in the MFMA register layout it is 4 registers per n-tile, and the add leaves the epilogue
    bool lean = false;
    if constexpr (sizeof(T) == 2 && !SPL) {
        const bool res_epi = p.epi == GP_EPI_SCALE_RES || p.epi == GP_EPI_RES_RELU;
        lean = !p.out_f32 && p.dbg != 5 && m0 + BM <= p.M && n0 + BN <= p.N && p.ldc % 8 == 0 &&
               ((size_t)p.C & 15) == 0 && !(res_epi && (p.gn_partial || p.ldres % 8 || ((size_t)p.res & 15)));
    }
    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        f32x4 init = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lean && p.bias && p.epi != GP_EPI_LNFOLD_GELU) init = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * NT * 16 + a * 16 + (lane >> 4) * 4);
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = init;
    }
    // split-operand mode: the cross terms (segments 0, 1) are accumulated first and scaled by 2^-S once, exactly, when the
    // hi x hi segment begins (a K range of a split-K launch that ends before that point scales after its loop)
    auto split_rescale = [&](int kt) {
        if constexpr (SPL) {
            if (kt + kt0 == 2 * p.split_n1) {
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < MT; ++b) acc[a][b] *= p.split_scale;
            }
        }
    };

    const int fr = lane & 15, fq = lane >> 4, sw = fr & 7;
    const int xfo = (wm * MT * 16 + fr) * RB, wfo = (wn * NT * 16 + fr) * RB;
    // DB: fragments of K sub-step 1 are fetched from LDS while the MFMAs of sub-step 0 run (explicit second
    // register set) and the MFMA cluster is bracketed by s_setprio -- an A/B arm, see scripts/gemm_bench.py.
    auto compute = [&](int buf) {
        const char* xs = smem + buf * STAGE + xfo;
        const char* ws = smem + buf * STAGE + BM * RB + wfo;
        if constexpr (RB == 64) {
            const int co = (fq ^ ((-(fr >> 2)) & 3)) << 4;
            uint4 xf[MT], wf[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const uint4*>(ws + t * 1024 + co);
#pragma unroll
            for (int t = 0; t < MT; ++t) xf[t] = *reinterpret_cast<const uint4*>(xs + t * 1024 + co);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma<T>(acc[nt][mt], wf[nt], xf[mt]);
        } else if constexpr (DB) {
            const int co0 = ((0 * 4 + fq) ^ sw) << 4, co1 = ((1 * 4 + fq) ^ sw) << 4;
            uint4 xf0[MT], wf0[NT], xf1[MT], wf1[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) wf0[t] = *reinterpret_cast<const uint4*>(ws + t * 2048 + co0);
#pragma unroll
            for (int t = 0; t < MT; ++t) xf0[t] = *reinterpret_cast<const uint4*>(xs + t * 2048 + co0);
#pragma unroll
            for (int t = 0; t < NT; ++t) wf1[t] = *reinterpret_cast<const uint4*>(ws + t * 2048 + co1);
#pragma unroll
            for (int t = 0; t < MT; ++t) xf1[t] = *reinterpret_cast<const uint4*>(xs + t * 2048 + co1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma<T>(acc[nt][mt], wf0[nt], xf0[mt]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma<T>(acc[nt][mt], wf1[nt], xf1[mt]);
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int co = ((ks * 4 + fq) ^ sw) << 4;
                uint4 xf[MT], wf[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const uint4*>(ws + t * 2048 + co);
#pragma unroll
                for (int t = 0; t < MT; ++t) xf[t] = *reinterpret_cast<const uint4*>(xs + t * 2048 + co);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma<T>(acc[nt][mt], wf[nt], xf[mt]);
            }
        }
    };

    constexpr int G = XI + WI;
    if constexpr (PP) {
        // ---- Ping-pong schedule (8 waves = two groups of four, one wave of each group per SIMD).  Per K step a wave
        // runs a LOAD phase [issue its 12 fragment reads of step t | issue the DMA of step t+LEAD | wait for its own
        // DMA of step t+1 | barrier] and an MFMA phase [wait for the fragments | 32 MFMAs | barrier]; group 1 starts
        // one barrier late, so on every SIMD one wave multiplies while its partner loads and the matrix pipe never
        // waits for an LDS read, a DMA issue or a barrier.  Ring of NS stages, DMA lead LEAD = NS - 2 steps, counted vmcnt.
        //   RAW: step t is read in interval 2t (group 0) / 2t+1 (group 1); every wave waited for its own DMA of step t
        //        in its load phase of step t-1 (intervals 2t-2 / 2t-1), i.e. before barrier 2t.
        //   WAR: the DMA of step t+LEAD overwrites the slot of step t-2 and is issued in interval 2t at the earliest;
        //        the last reads of step t-2 (group 1, issued in interval 2t-3) were retired by the lgkmcnt(0) at the
        //        head of interval 2t-2: two barriers earlier.
        constexpr int LEAD = NS - 2;
        const int grp = wave >> 2;
        const int co = (fq ^ ((-(fr >> 2)) & 3)) << 4;
        uint4 xf[MT], wf[NT];
        // plain GEMM: scalar-base DMA.  Rows past the M / N edge are clamped to the last valid row (their products are
        // never stored), so no zero page and no per-lane select is needed.
        const bool sdma = !p.conv && (long)BM * p.ldx * sizeof(T) < (1l << 31) && (long)BN * p.K * sizeof(T) < (1l << 31);
        unsigned xo[XI], wo[WI];
#pragma unroll
        for (int i = 0; i < XI; ++i) {
            const int r = min((i * NW + wave) * RPI + lrow, p.M - 1 - m0);
            xo[i] = (unsigned)(((long)r * p.ldx + lchunk * EPT) * sizeof(T));
        }
#pragma unroll
        for (int i = 0; i < WI; ++i) {
            const int r = min((i * NW + wave) * RPI + lrow, p.N - 1 - n0);
            wo[i] = (unsigned)(((long)r * p.K + lchunk * EPT) * sizeof(T));
        }
        // (round 5: issuing the W pieces of a step's DMA between the MFMAs of the MFMA phase instead of in the load phase -- a PPSPLIT instantiation, no
        // branch in the MFMA stream -- changed nothing: profiles/r05_pp_split_dma_ab.txt; which wave of a SIMD issues them does not matter)
        // (round 5: addressing both operands K-blocked -- every LDS-DMA instruction 1 KB contiguous -- was timed here and bought 2-4 %:
        // profiles/r05_kblock_probe.txt; the arms are gone again)
        const char* xtile = reinterpret_cast<const char*>(X + (long)m0 * p.ldx);
        const char* wtile = reinterpret_cast<const char*>(W + (long)n0 * p.K);
        auto stage_s = [&](int buf, int kt) {
            const unsigned xs = lds0 + buf * STAGE + wave * 1024;
            const unsigned ws = xs + BM * RB;
            long xadd = 0, wadd = 0;
            if constexpr (SPL) {
                const int seg = (kt >= p.split_n1 ? 1 : 0) + (kt >= 2 * p.split_n1 ? 1 : 0);
                kt -= seg * p.split_n1;
                xadd = seg == 1 ? p.xplane_b : 0;
                wadd = seg == 0 ? p.wplane_b : 0;
            }
            const char* bx = xtile + (long)kt * RB + xadd;
            const char* bw = wtile + (long)kt * RB + wadd;
#pragma unroll
            for (int i = 0; i < XI; ++i) glds16_s(bx, xo[i], xs + i * NW * 1024);
#pragma unroll
            for (int i = 0; i < WI; ++i) glds16_s(bw, wo[i], ws + i * NW * 1024);
        };
#pragma unroll
        for (int i = 0; i < LEAD; ++i)
            if (i < p.nkt) { if (sdma) stage_s(i, i); else stage(i, i); }
        // wait for step 0 (the load phase of step t waits for step t+1)
        if constexpr (LEAD == 3) {
            if (p.nkt >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
            else if (p.nkt == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (p.nkt >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        prefetch_retire(pfs);
        __builtin_amdgcn_s_barrier();
        if (grp) __builtin_amdgcn_s_barrier();
        int buf = 0, nbuf = LEAD;
        clk_stamp(p, 0);
        // GP_PP_STAMPS (investigation build, scripts/pp_stamps.py, profiles/r05_pp_stamps.txt): s_memtime stamps of waves 0 and 4 of workgroup 0 inside the steps
        // 8 .. 23 of the main loop -- 0 phase start, 1 fragment reads issued, 2 DMA issued, 3 own DMA landed (vmcnt), 4 behind the barrier, 5 fragments there
        // (lgkmcnt), 6 MFMAs issued, 7 behind the second barrier -- into the workspace behind the clock stamps
#ifdef GP_PP_STAMPS
#define GP_PPS(k)                                                                                                              \
        do {                                                                                                                     \
            if (p.ws && p.splitk <= 1 && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && kt >= 8 && kt < 24) {                  \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
                unsigned long long t_;                                                                                           \
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory");                                    \
                reinterpret_cast<unsigned long long*>(p.ws)[4096 + ((wave >> 2) * 16 + (kt - 8)) * 8 + (k)] = t_;                 \
                __builtin_amdgcn_sched_barrier(0);                                                                               \
            }                                                                                                                    \
        } while (0)
#else
#define GP_PPS(k) do {} while (0)
#endif
        for (int kt = 0; kt < p.nkt; ++kt) {
            GP_PPS(0);
            if (p.dbg != 2) {
                const char* xs = smem + buf * STAGE + xfo;
                const char* ws = smem + buf * STAGE + BM * RB + wfo;
#pragma unroll
                for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const uint4*>(ws + t * 1024 + co);
#pragma unroll
                for (int t = 0; t < MT; ++t) xf[t] = *reinterpret_cast<const uint4*>(xs + t * 1024 + co);
            }
            __builtin_amdgcn_sched_barrier(0);
            GP_PPS(1);
            if (kt + LEAD < p.nkt && (p.dbg != 1 || kt + LEAD < NS)) { if (sdma) stage_s(nbuf, kt + LEAD); else stage(nbuf, kt + LEAD); }
            GP_PPS(2);
            // own DMA of step t+1 landed; steps t+2 .. t+LEAD stay in flight
            if (LEAD == 3 && kt + 3 < p.nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
            else if (kt + 2 < p.nkt) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            GP_PPS(3);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            GP_PPS(4);
#ifndef GP_PP_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
            GP_PPS(5);
            __builtin_amdgcn_sched_barrier(0);
            split_rescale(kt);
            __builtin_amdgcn_s_setprio(1);
            if (p.dbg != 2) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) mma<T>(acc[nt][mt], wf[nt], xf[mt]);
            }
            __builtin_amdgcn_s_setprio(0);
            GP_PPS(6);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            GP_PPS(7);
            __builtin_amdgcn_sched_barrier(0);
            buf = buf + 1 == NS ? 0 : buf + 1;
            nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
        }
        clk_stamp(p, 1);
        if (!grp) __builtin_amdgcn_s_barrier();
    } else
    if constexpr (DB && NS == 2) {
        // Software-pipelined schedule (2 LDS stages, 2 fragment register sets F0/F1), per K step t:
        //   read F1 <- (t, k-half 1) | MFMA(F0) | wait own LDS reads + own DMA(t+1), barrier |
        //   DMA(t+2) into the buffer just freed | read F0 <- (t+1, k-half 0) | MFMA(F1)
        // so every LDS fragment read and the DMA issue run under the MFMAs of the other half step instead of
        // all eight waves hitting the LDS together right after the barrier.
        uint4 xf0[MT], wf0[NT], xf1[MT], wf1[NT];
        const int co0 = ((0 * 4 + fq) ^ sw) << 4, co1 = ((1 * 4 + fq) ^ sw) << 4;
        auto rd = [&](int buf, int co, uint4* xf, uint4* wf) {
            const char* xs = smem + buf * STAGE + xfo;
            const char* ws = smem + buf * STAGE + BM * RB + wfo;
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const uint4*>(ws + t * 2048 + co);
#pragma unroll
            for (int t = 0; t < MT; ++t) xf[t] = *reinterpret_cast<const uint4*>(xs + t * 2048 + co);
        };
        auto mm = [&](const uint4* xf, const uint4* wf) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma<T>(acc[nt][mt], wf[nt], xf[mt]);
        };
        stage(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        prefetch_retire(pfs);
        __syncthreads();
        rd(0, co0, xf0, wf0);
        if (p.nkt > 1) stage(1, 1);
        for (int kt = 0; kt < p.nkt; ++kt) {
            const int buf = kt & 1;
            split_rescale(kt);
            rd(buf, co1, xf1, wf1);
            __builtin_amdgcn_sched_barrier(0);
            mm(xf0, wf0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < p.nkt) stage(buf, kt + 2);
            if (kt + 1 < p.nkt) rd(buf ^ 1, co0, xf0, wf0);
            __builtin_amdgcn_sched_barrier(0);
            mm(xf1, wf1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    } else {
    // NS-stage ring: steps t+1 .. t+NS-1 are in flight while step t is multiplied.  Each wave waits for its own
    // DMAs of step t+1 with a COUNTED vmcnt (the NS-2 younger steps stay in flight across the barrier).
#pragma unroll
    for (int i = 0; i < NS - 1; ++i)
        if (i < p.nkt) stage(i, i);
    if (p.nkt > NS - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    prefetch_retire(pfs);
    __syncthreads();
    int buf = 0, nbuf = NS - 1;
    for (int kt = 0; kt < p.nkt; ++kt) {
        const bool more = kt + NS - 1 < p.nkt;
        if (more && p.dbg != 1) stage(nbuf, kt + NS - 1);
        split_rescale(kt);
        if (p.dbg != 2) compute(buf);
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G * (NS - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }
    }

    if constexpr (SPL) {
        if (kt0 + p.nkt <= 2 * p.split_n1) {
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) acc[a][b] *= p.split_scale;
        }
    }
    if (p.dbg == 3) {   // timing ablation: no epilogue (keeps the accumulators alive, stores nothing in practice)
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < NT; ++a)
#pragma unroll
            for (int b = 0; b < MT; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
        if (s == 12345.678f) reinterpret_cast<float*>(p.C)[0] = s;
        return;
    }
    if constexpr (sizeof(T) == 2) {
        if (lean) {
            static_assert(NS * STAGE >= NW * 9216, "lean epilogue slabs must fit");
            char* slab = smem + wave * 9216;
            const int mb = m0 + wm * MT * 16, nb = n0 + wn * NT * 16;
            switch (p.epi) {
                case GP_EPI_GELU: epilogue_lean<MT, NT, GP_EPI_GELU>(p, acc, slab, mb, nb, lane); break;
                case GP_EPI_RELU: epilogue_lean<MT, NT, GP_EPI_RELU>(p, acc, slab, mb, nb, lane); break;
                case GP_EPI_LRELU: epilogue_lean<MT, NT, GP_EPI_LRELU>(p, acc, slab, mb, nb, lane); break;
                case GP_EPI_SCALE_RES: epilogue_lean<MT, NT, GP_EPI_SCALE_RES>(p, acc, slab, mb, nb, lane); break;
                case GP_EPI_RES_RELU: epilogue_lean<MT, NT, GP_EPI_RES_RELU>(p, acc, slab, mb, nb, lane); break;
                case GP_EPI_LNFOLD_GELU: epilogue_lean<MT, NT, GP_EPI_LNFOLD_GELU>(p, acc, slab, mb, nb, lane); break;
                default: epilogue_lean<MT, NT, GP_EPI_NONE>(p, acc, slab, mb, nb, lane); break;
            }
            return;
        }
    }
    if constexpr (SPL) epilogue_split<MT, NT>(p, acc, smem + wave * 8192, m0 + wm * MT * 16, n0 + wn * NT * 16, lane);
    else epilogue_generic<RT, MT, NT, SPL>(p, acc, smem + wave * 8192, m0 + wm * MT * 16, n0 + wn * NT * 16, lane);
}

// =====================================================================================================
// 3x3 / stride 1 / pad 1 convolution with Cout = 256 on the ping-pong schedule, X staged as an LDS WINDOW.
// The implicit-GEMM kernels above fetch the X rows of every filter tap separately: 9 x 16 KB of LDS-DMA per 32-channel
// chunk and tile, although the nine taps read the same (TR + 2) x (W + 2) pixel window shifted by (kh, kw).  The
// per-CU LDS-DMA rate (~53 GB/s) is what bounds the 256 x 256 tile, so here the K loop runs channel chunk OUTER /
// tap INNER: the window of a chunk (<= 25 KB, halo and image border from the zero page) is fetched ONCE, double
// buffered against the previous chunk, and the nine taps read their B fragments from it at shifted pixel positions
// (16 consecutive window pixels per fragment; bit 1 of the 16-byte channel slot is XOR-ed with bit 2 of the window pixel
// index: the one swizzle family -- found by enumeration over the ds_read_b128 lane groups {0-3,12-15,20-27}, ... --
// that keeps every lane group on 16 different slots for EVERY pixel alignment, i.e. for all nine tap shifts).  W still streams tap by tap through
// the 4-stage ring.  DMA bytes per tile: 8 x (25 + 144) KB instead of 8 x 288 KB.
// Tile = 256 output pixels = TR whole image rows (W in {64, 32, 16}); 8 waves, two ping-pong groups as above.
__device__ __forceinline__ void wait_vmcnt(int n) {   // s_waitcnt takes an immediate: wave-uniform dispatch on n
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void wait_vmcnt_n(int n) {   // as wait_vmcnt for counts up to 24 (larger n waits for 24: stricter)
#define GP_WV(i) case i: asm volatile("s_waitcnt vmcnt(" #i ")" ::: "memory"); break;
    switch (n) {
        GP_WV(0) GP_WV(1) GP_WV(2) GP_WV(3) GP_WV(4) GP_WV(5) GP_WV(6) GP_WV(7) GP_WV(8) GP_WV(9) GP_WV(10) GP_WV(11) GP_WV(12)
        GP_WV(13) GP_WV(14) GP_WV(15) GP_WV(16) GP_WV(17) GP_WV(18) GP_WV(19) GP_WV(20) GP_WV(21) GP_WV(22) GP_WV(23)
        default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
    }
#undef GP_WV
}

// SPL (split-operand mode, GemmKP::split_n1): the (chunk, tap) loop runs three times -- X_hi windows against W_lo' rows, X_lo'
// against W_hi, then (accumulators scaled by 2^-S) X_hi against W_hi -- as one stream of 3 * NCC chunks through the same W
// ring and window double buffer; fp32 output through the generic epilogue.
// WIDE (round 4): a tile of 512 pixels x 128 output channels instead of 256 x 256 -- the same MFMAs, fragment reads and LDS per
// workgroup, but per K step ONE W piece per wave instead of two and 1 MB instead of 1.37 MB of LDS-DMA per tile (the W rows of a
// step are 8 KB, the window of a chunk 45 KB): the load phase of the ping-pong schedule, which is what bounds the kernel, carries
// a third fewer LDS-DMA instructions.  Two workgroups (the two channel halves) share a pixel tile's window through the XCD's L2.
template <int WIMG, int NS, bool SPL = false, bool GNL = false, bool WIDE = false>    // GNL: measurement arm (GroupNorm-apply + GELU in the loader), see below
__global__ __launch_bounds__(512) void conv3_pp_kernel(const GemmKP p) {
    constexpr int MT = 8, NT = 4, BM = WIDE ? 512 : 256, BN = WIDE ? 128 : 256, LEAD = NS - 2;
    constexpr int NTN = 256 / BN, WPI = BN / 128;       // channel tiles per pixel tile; W pieces (1 KB) per wave and K step
    static_assert(!WIDE || (WIMG >= 32 && !SPL), "wide tile: whole rows of ONE image (W >= 32), fp16 mode");
    // window row pitch WW: a multiple of 8 pixels, so that a kh shift never changes bit 2 of the window pixel index
    // (the swizzle bit) and the nine tap addresses of an m-tile are 3 registers (one per kw) + an immediate offset
    constexpr int TR = BM / WIMG, WW = (WIMG + 2 + 7) / 8 * 8, NP = (TR + 2) * WW, NI = (NP + 15) / 16;
    constexpr int WINB = WIDE ? 49152 : 32768, WST = BN * 64;     // two window buffers
    constexpr int XJ = (NI + 7) / 8;                // window pieces per wave and chunk
    static_assert(NI * 1024 <= WINB, "window");
    constexpr int WIN0 = NS * WST;
    constexpr int SMEM = WIN0 + 2 * WINB;
    static_assert(SMEM >= 8 * 9216 && SMEM <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WIDE ? (wave & 3) : (wave & 1), wn = WIDE ? (wave >> 2) : (wave >> 1), grp = wave >> 2;
    const int fr = lane & 15, fq = lane >> 4;

    const int nblk = p.tiles_m * NTN;
    const int bid = blockIdx.x;
    const int q8 = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    const int m0 = (tile / NTN) * BM, n0 = (tile % NTN) * BN;
    const int img = m0 / (p.H * WIMG), h0 = (m0 - img * p.H * WIMG) / WIMG;
    const int Cin = p.Cin, NCC = Cin >> 5, NG = SPL ? 3 * NCC : NCC, NSTG = 9 * NG;   // NG chunks of 32 channels in all

    const half_t* __restrict__ X = reinterpret_cast<const half_t*>(p.X);
    const half_t* __restrict__ W = reinterpret_cast<const half_t*>(p.W);
    const char* zp = reinterpret_cast<const char*>(gp_zero_page_tu);

    // ---- DMA sources
    const int lrow = lane >> 2;
    const int wchunk = (lane & 3) ^ ((-(lrow >> 2)) & 3);
    unsigned woff[WPI];        // per-lane byte offsets inside W; the K step's offset goes into the scalar base
#pragma unroll
    for (int i = 0; i < WPI; ++i) woff[i] = (unsigned)(((n0 + (i * 8 + wave) * 16 + lrow) * p.K + wchunk * 8) * 2);
    // window DMA instruction wave + 8 j: pixel 16 i + lane / 4, physical chunk lane & 3.  32-bit byte offsets from the
    // image base (scalar); halo / border lanes point at the zero page instead (xvalid)
    unsigned xoff[XJ];
    unsigned xvalid = 0;
    const char* ximg = reinterpret_cast<const char*>(X + (long)img * p.H * WIMG * Cin);
    const PfSink pfs = prefetch_issue(p, blockIdx.x, gridDim.x, wave, 8, lane);   // gp_gemm_desc.prefetch (hint)
#pragma unroll
    for (int j = 0; j < XJ; ++j) {
        const int i = wave + 8 * j, px = i * 16 + lrow;
        const int wy = px / WW, wx = px - wy * WW;
        const int gy = h0 - 1 + wy, gx = wx - 1;
        const int lc = (lane & 3) ^ (((px >> 2) & 1) << 1);
        const bool ok = i < NI && px < NP && wx < WIMG + 2 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)WIMG;
        xoff[j] = ok ? (unsigned)(((gy * WIMG + gx) * Cin + lc * 8) * 2) : 0u;
        if (ok) xvalid |= 1u << j;
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;
    auto stage_w = [&](int buf, int g, int tap) {      // W rows of K step (chunk g, tap): k offset tap * Cin + cc * 32
        int cc = g;
        long plane = 0;
        if constexpr (SPL) { const int seg = (g >= NCC ? 1 : 0) + (g >= 2 * NCC ? 1 : 0); cc = g - seg * NCC; plane = seg == 0 ? p.wplane_b : 0; }
        const char* base = reinterpret_cast<const char*>(W) + plane + (tap * Cin + cc * 32) * 2;
        const unsigned d = lds0 + buf * WST + wave * 1024;
        glds16_s(base, woff[0], d);
        if constexpr (WPI == 2) glds16_s(base, woff[WPI - 1], d + 8192);
    };
    auto stage_x = [&](int j, int g) {        // one 16-pixel piece of the window of chunk g
        int cc = g;
        long plane = 0;
        if constexpr (SPL) { const int seg = (g >= NCC ? 1 : 0) + (g >= 2 * NCC ? 1 : 0); cc = g - seg * NCC; plane = seg == 1 ? p.xplane_b : 0; }
        const unsigned d = lds0 + WIN0 + (g & 1) * WINB + (wave + 8 * j) * 1024;
        const char* src = ximg + plane + xoff[j] + cc * 64;
        glds16((xvalid >> j) & 1 ? src : zp, d);
    };

    // ---- accumulators (bias as the initial value: the lean epilogue is the only one here)
    f32x4 acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        f32x4 init = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!SPL && p.bias) init = *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 64 + a * 16 + fq * 4);   // (SPL: the generic epilogue adds it)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = init;
    }
    // LDS byte address of this lane's B fragment per m-tile for kw = 0; tap (kh, kw) adds the lane constant xd[kw] and
    // kh * WW * 64 as an immediate.  xd does not depend on the m-tile: the window pixel index of (mt, fr) is
    // pb = (ml / WIMG) * WW + ml % WIMG with ml = wm * 128 + mt * 16 + fr, and WW, 16 and WIMG are multiples of 8, so
    // pb % 8 == fr % 8 for every mt -- and the swizzle term only looks at bit 2 of the pixel index.  (One address per
    // (mt, kw) cost 24 registers and made hipcc spill: scratch is banned on the path)
    unsigned xa[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int ml = wm * 128 + mt * 16 + fr;
        const int pb = (ml / WIMG) * WW + (ml % WIMG);
        xa[mt] = WIN0 + pb * 64 + ((fq ^ (((pb >> 2) & 1) << 1)) << 4);   // offset into smem
    }
    static_assert(WW % 8 == 0 && WIMG % 8 == 0, "the kw deltas must not depend on the m-tile");
    unsigned xd[3];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const int p0 = fr & 7, p1 = p0 + kw;
        xd[kw] = kw * 64 + (((fq ^ (((p1 >> 2) & 1) << 1)) - (fq ^ (((p0 >> 2) & 1) << 1))) << 4);
    }
    const int wfo = (wn * 64 + fr) * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);

    // ---- prologue: window of chunk 0, W of steps 0 and 1
#pragma unroll
    for (int j = 0; j < XJ; ++j)
        if (wave + 8 * j < NI) stage_x(j, 0);
#pragma unroll
    for (int i = 0; i < LEAD; ++i) stage_w(i, 0, i);
    wait_vmcnt(WPI * (LEAD - 1));
    prefetch_retire(pfs);
    __builtin_amdgcn_s_barrier();
    if (grp) __builtin_amdgcn_s_barrier();

    uint4 xf[MT], wf[NT];
    clk_stamp(p, 0);
    int st = 0, rbuf = 0, wbuf = LEAD, ops1 = WPI, ops2 = WPI;   // DMA ops issued one / two phases ago (W of steps 1..LEAD-1 at first)
    for (int cc = 0; cc < NG; ++cc) {      // (cc = global chunk index; SPL: segment cc / NCC)
        if constexpr (SPL) {
            if (cc == 2 * NCC) {            // the cross terms are complete: scale them once, exactly, then add x_hi w_hi
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < MT; ++b) acc[a][b] *= p.split_scale;
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap, ++st) {
            const int kh = tap / 3, kw = tap - kh * 3;
            const char* ws = smem + rbuf * WST + wfo;
#pragma unroll
            for (int t = 0; t < NT; ++t) wf[t] = *reinterpret_cast<const uint4*>(ws + t * 1024);
            unsigned xdk = xd[kw];
            asm volatile("" : "+v"(xdk));   // opaque per tap: keeps hipcc from hoisting xa + xd into 24 loop-invariant registers
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                xf[mt] = *reinterpret_cast<const uint4*>(smem + (xa[mt] + xdk) + kh * WW * 64);
            __builtin_amdgcn_sched_barrier(0);
            // DMA: one window piece of the NEXT chunk during taps 1..XJ (its buffer was last read in chunk cc - 1, whose
            // reads are two barriers back by tap 1), then the W rows of step st + 2; then wait for the own W of st + 1
            bool xi = false;
            if (tap >= 1 && tap <= XJ) {
                xi = cc + 1 < NG && wave + 8 * (tap - 1) < NI && p.dbg != 1;
                if (xi) stage_x(tap - 1, cc + 1);
            }
            const bool wi = st + LEAD < NSTG && p.dbg != 1;
            if (wi) stage_w(wbuf, cc + (tap + LEAD) / 9, (tap + LEAD) % 9);
            if (GNL && tap >= 2 && tap <= 5) {
                // MEASUREMENT ARM (wrong results; scripts/conv_gn_loader_ab.py, profiles/r04_conv_gn_loader_ab.txt): what GroupNorm-apply +
                // GELU inside the window loader would cost at the least.  The window piece this wave requested at tap - 1 has landed (the
                // counted wait of the previous load phase covers it): one pass over the wave's OWN piece in LDS -- read 16 bytes per lane,
                // scale / shift / GELU eight values, write back -- in the load phase, where it would have to live.  (A real version adds
                // the per-(image, channel) table look-up and the border mask on top.)
                const int j = tap - 2;
                if (cc + 1 < NG && wave + 8 * j < NI) {
                    char* a = smem + WIN0 + ((cc + 1) & 1) * WINB + (wave + 8 * j) * 1024 + lane * 16;
                    half8 h = *reinterpret_cast<half8*>(a);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {      // one 2-vector at a time: the kernel has no registers to spare
                        f32x2 f = f32x2{fmaf((float)h[2 * e], p.split_scale, 0.25f), fmaf((float)h[2 * e + 1], p.split_scale, 0.25f)};
                        f = gelu_poly2(f);
                        h[2 * e] = (half_t)f[0]; h[2 * e + 1] = (half_t)f[1];
                    }
                    *reinterpret_cast<half8*>(a) = h;
                }
            }
            // the own W rows of step st + 1 have landed once at most the ops issued after them are outstanding: those of
            // this phase and (LEAD = 3) of the previous one (vmcnt retires in issue order, the window pieces included)
            const int ops0 = (wi ? WPI : 0) + (xi ? 1 : 0);
            wait_vmcnt(LEAD == 3 ? ops0 + ops1 : ops0);
            ops2 = ops1; ops1 = ops0;
            rbuf = rbuf + 1 == NS ? 0 : rbuf + 1;
            wbuf = wbuf + 1 == NS ? 0 : wbuf + 1;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) mma<half_t>(acc[nt][mt], wf[nt], xf[mt]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) xa[mt] += (cc & 1) ? -WINB : WINB;     // the other window buffer
    }
    clk_stamp(p, 1);
    if (!grp) __builtin_amdgcn_s_barrier();

    const int mb = m0 + wm * 128, nb = n0 + wn * 64;
    if constexpr (SPL) {
        epilogue_split<MT, NT>(p, acc, smem + wave * 8192, mb, nb, lane);
    } else {
        char* slab = smem + wave * 9216;
        switch (p.epi) {
            case GP_EPI_GELU: epilogue_lean<MT, NT, GP_EPI_GELU>(p, acc, slab, mb, nb, lane); break;
            case GP_EPI_RELU: epilogue_lean<MT, NT, GP_EPI_RELU>(p, acc, slab, mb, nb, lane); break;
            default: epilogue_lean<MT, NT, GP_EPI_NONE>(p, acc, slab, mb, nb, lane); break;
        }
    }
}

// =====================================================================================================
// Weights-in-registers GEMM for K = 512 (variant 16): C[M][N] = act(X[M][512] . W[N][512]^T + bias), fp16.
// The tile kernels above move every operand byte through an LDS ring, and what a CU can pull from L2 (~53-70 GB/s) is
// their wall: stage-2 fc1 (M 16384, N 2048, K 512) needs 1.57 MB per CU in 256 x 128 tiles.  With K = 512 a slice of
// 256 output channels of W is 256 KB -- half of a CU's vector register file.  So a workgroup (8 waves, one per CU)
// keeps ITS 256 rows of W in registers for the whole launch (a wave owns 32 rows = 32 A fragments = 128 VGPRs, loaded
// once, straight from global memory in the MFMA layout) and streams a group of X rows past them: X tiles of 32 rows x
// 1 KB go through a 4-buffer LDS ring by LDS-DMA (one instruction = one row, chunk-swizzled on the source side), every
// wave reads the whole tile as B fragments (2 ds_read_b128 per 4 MFMAs).  Per CU: 256 KB + rows x 1 KB (0.77 MB at
// 512 rows per group) and ONE barrier per 32 rows instead of one per K step.
// Schedule (one period = one tile, one barrier): every wave runs the 64 MFMAs of tile q with the ACTIVATION of tile
// q-1 in their shadow -- after each MFMA four plain v_fma_f32 / v_mul_f32 of the GELU polynomial (12 slices of one
// operation per element, two chains per MFMA; inline asm, not packed: beside MFMAs a v_pk_fma_f32 costs ~22 cycles
// more than two v_fma_f32) -- B fragments read two K steps ahead with counted lgkmcnt, the wave's four LDS-DMA
// instructions of tile q+3 spread over the K steps (the two waves of a SIMD issue theirs two K steps apart), then the
// four 8-byte stores of tile q-1.  sched_barriers pin that order (hipcc otherwise clusters the VALU work and waits
// lgkmcnt(0) in front of every MFMA pair; sched_group_barrier pipelines did the same).
// What bounds it (s_memtime stamps, scripts/wreg_stamps.py): the SIMD's vector ISSUE.  Per period and SIMD 128 MFMAs hold
// the issue port 8 of their 16 cycles (1 024) and the GELU of 2 x 16 x 64 elements is 472 four-cycle instructions
// (1 888): the MFMA + GELU block takes 3 400-3 600 of the period's ~4 100 cycles (the rest: stores, vmcnt wait,
// barrier), so the matrix pipe cannot be more than ~58 % busy while the GELU shares the SIMD.  Measured (interleaved
// medians, scripts/gemm_ab.py): 48.3 us against 47.5 (ping-pong tile) and 53.4 (256 x 128 tile) alone; end to end the
// step gains 4 % serial and 0.7 % with three batches in flight, because a workgroup's H rows stay in the L2 of the XCD
// that the following fc2 launch reads them from (fc2 49.6 -> 45.3 us).
// Hazards: buffer (q+3)&3 was last read in period q-1, which every wave left through the barrier that opens period q
// (WAR); a wave waits for ITS four DMA instructions of tile q+1 with a counted vmcnt right before the barrier that
// closes period q, the barrier publishes all 32 rows (RAW).  vmcnt counts in issue order, loads and stores alike, so
// the count is the number of vector-memory instructions the wave issued after the DMA of tile q+1: the DMA of tiles
// q+2, q+3 and the stores of periods q-2, q-1, q (20 in the steady state) -- every store is ONE inline-asm instruction so
// that the count is not the compiler's to change (nst() / ndma() below).
template <int EPI, int ABL = 0>   // ABL: investigation builds: 1 no MFMA, 2 no stores, 4 no in-loop DMA, 8 no GELU shadow (wrong results, not
                                 // instantiated), 16 time stamps (right results)
__global__ __launch_bounds__(512) void gemm_wreg_kernel(const GemmKP p) {
    constexpr int K = 512, KS = K / 32, TM = 32, NBUF = 4, BUFB = TM * K * 2, NT = 2, MT = 2;
    static_assert(NT * MT * 2 == 8, "the epilogue walks eight 2-vectors");
    __shared__ __attribute__((aligned(1024))) char smem[NBUF * BUFB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    // work item: all 8 N slices of an M group on one XCD (the X rows of the group enter one L2 only)
    const int nsl = p.N >> 8;
    const int bid = blockIdx.x;
    int item;
    {
        const int per = gridDim.x >> 3;                          // host: gridDim.x % 8 == 0
        item = (bid & 7) * per + (bid >> 3);                     // XCD-contiguous chunks of the (m group, n slice) list
    }
    const int mg = item / nsl, ns = item - mg * nsl;
    const long m0 = (long)mg * p.tiles_m * TM;                   // tiles_m = tiles per M group
    const int T = min(p.tiles_m, (int)((p.M - m0) / TM));
    if (T <= 0) return;
    const int nb = ns * 256 + wave * 32;

    // ---- DMA: wave w fetches rows 4w .. 4w+3 of a tile; lane c fetches the 16-byte chunk c ^ (row & 15) of the row
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;
    const char* xsrc = reinterpret_cast<const char*>(p.X) + ((m0 + wave * 4) * (long)p.ldx) * 2;
    const long rowb = (long)p.ldx * 2;
    unsigned xoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xoff[i] = (unsigned)(i * rowb) + ((lane ^ ((wave * 4 + i) & 15)) << 4);
    auto dma = [&](int t) {
        const char* src = xsrc + (long)t * TM * rowb;
        const unsigned dst = lds0 + (t & (NBUF - 1)) * BUFB + wave * 4 * 1024;
#pragma unroll
        for (int i = 0; i < 4; ++i) glds16_s(src, xoff[i], dst + i * 1024);
    };
    // B fragment (mt, ks) of a tile: row mt*16 + fr, chunk (ks*4 + fq) ^ fr = ((ks & 3)*4 ^ (fq ^ fr)) + (ks >> 2) * 16
    unsigned boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = fr * 1024 + (((j * 4) ^ (fq ^ fr)) << 4);

    uint4 wf[NT][KS];   // the resident W fragments (loaded below, after the first X tiles have been requested)
    f32x4 b4[NT], acc[NT][MT];
    f32x2 v[NT * MT * 2];   // pre-activation values of the previous tile
    half_t* Cw = reinterpret_cast<half_t*>(p.C) + (m0 + fr) * (long)p.ldc + nb + fq * 4;
    // B fragments of a tile are read two K steps ahead of their MFMAs into a ring of three register pairs; SHADOW: the
    // GELU of the previous tile (v) is issued in the shadow of the MFMAs, slices of plain VALU operations behind each MFMA
    // (12 slices x 4 chains per half tile: K steps 0-7 carry chains 0-3, K steps 8-15 chains 4-7).  sched_barriers pin
    // the order: hipcc otherwise clusters the VALU work and waits lgkmcnt(0) in front of every MFMA pair.
    f32x2 gx[4], gt[4], gp[4];
    float c1v = GELU_H[1];
    asm volatile("" : "+v"(c1v));   // a VGPR constant
    auto mfmas = [&](int t, auto shadow, int tdma) {   // tdma: tile whose four DMA instructions ride along (< 0: none)
        constexpr bool SH = decltype(shadow)::value && EPI == GP_EPI_GELU && !(ABL & 8);
        const char* dsrc = xsrc + (long)tdma * TM * rowb;
        const unsigned ddst = lds0 + (tdma & (NBUF - 1)) * BUFB + wave * 4 * 1024;
        const char* xb = smem + (t & (NBUF - 1)) * BUFB;
        uint4 bf[3][MT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                bf[ks][mt] = *reinterpret_cast<const uint4*>(xb + boff[ks & 3] + mt * 16 * 1024 + (ks >> 2) * 256);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks + 2 < KS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    bf[(ks + 2) % 3][mt] = *reinterpret_cast<const uint4*>(xb + boff[(ks + 2) & 3] + mt * 16 * 1024 + ((ks + 2) >> 2) * 256);
            }
            // the issuing wave stalls ~100 cycles per LDS-DMA instruction; the two waves of a SIMD (w, w + 4) issue theirs two
            // K steps apart so that one of them keeps the SIMD busy
            if constexpr (ks % 4 == 0) {
                if (tdma >= 0 && wave >= 4) glds16_s(dsrc, xoff[ks / 4], ddst + (ks / 4) * 1024);
            } else if constexpr (ks % 4 == 2) {
                if (tdma >= 0 && wave < 4) glds16_s(dsrc, xoff[ks / 4], ddst + (ks / 4) * 1024);
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NT * MT>([&](auto jc) {
                constexpr int j = decltype(jc)::value, nt = j / MT, mt = j % MT;
                if constexpr (!(ABL & 1)) mma<half_t>(acc[nt][mt], wf[nt][ks], bf[ks % 3][mt]);
                else acc[nt][mt][0] += __builtin_bit_cast(float, bf[ks % 3][mt].x ^ wf[nt][ks].x);
                if constexpr (SH) {
                    constexpr int half = ks / 8, slot = (ks % 8) * 2 + j / 2;      // slice index of this K step: two per K step
                    constexpr int c0 = (j % 2) * 2;                                 // chains c0, c0 + 1
                    if constexpr (slot < GELU_SLICES) {
                        gelu_poly2_slice<slot>(v[half * 4 + c0], gx[c0], gt[c0], gp[c0], c1v);
                        gelu_poly2_slice<slot>(v[half * 4 + c0 + 1], gx[c0 + 1], gt[c0 + 1], gp[c0 + 1], c1v);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };
    auto take = [&]() {   // accumulators -> v, accumulators back to the bias
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 a = acc[nt][mt];
                acc[nt][mt] = b4[nt];
                v[(mt * NT + nt) * 2] = f32x2{a[0], a[1]};
                v[(mt * NT + nt) * 2 + 1] = f32x2{a[2], a[3]};
            }
    };
    auto activate = [&]() {
        if constexpr (EPI == GP_EPI_GELU) {
            gelu_poly2_xn<4>(v);       // two rounds of four lock-step chains (register budget)
            gelu_poly2_xn<4>(v + 4);
        } else if constexpr (EPI == GP_EPI_RELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = f32x2{fmaxf(v[i][0], 0.0f), fmaxf(v[i][1], 0.0f)};
        } else if constexpr (EPI == GP_EPI_LRELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = f32x2{v[i][0] > 0.0f ? v[i][0] : 0.1f * v[i][0], v[i][1] > 0.0f ? v[i][1] : 0.1f * v[i][1]};
        }
    };
    auto stores = [&](int t) {   // NT*MT stores of 8 bytes per lane, one instruction each (counted below)
        half_t* c = Cw + (long)t * TM * p.ldc;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x2 lo = v[(mt * NT + nt) * 2], hi = v[(mt * NT + nt) * 2 + 1];
                half4 o = {(half_t)lo[0], (half_t)lo[1], (half_t)hi[0], (half_t)hi[1]};
                const half_t* dst = c + (long)mt * 16 * p.ldc + nt * 16;
                if constexpr (!(ABL & 2)) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(dst), "v"(o) : "memory");
                else asm volatile("" ::"v"(dst), "v"(o));
            }
    };
    // vector-memory instructions of this wave per period q: the DMA of tile q+3, then the stores of tile q-1
    auto ndma = [&](int t) { return t < T && (!(ABL & 4) || t < 3) ? 4 : 0; };
    auto nst = [&](int q) { return q >= 1 && !(ABL & 2) ? NT * MT : 0; };

#pragma unroll
    for (int t = 0; t < 3; ++t)
        if (t < T) dma(t);
    const PfSink pfs = prefetch_issue(p, bid, gridDim.x, wave, 8, lane);
    // ---- W fragments: lane (fr, fq) of fragment (nt, ks) holds W[nb + nt*16 + fr][ks*32 + fq*8 .. +8]
    const half_t* Wp = reinterpret_cast<const half_t*>(p.W) + (long)(nb + fr) * K + fq * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[nt][ks] = *reinterpret_cast<const uint4*>(Wp + (long)nt * 16 * K + ks * 32);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        b4[nt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + nt * 16 + fq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};

    // everything in flight so far (three X tiles, W, bias) has landed before the loop starts.  The builtin, not inline
    // asm: hipcc tracks it, and would otherwise wait for the W loads inside the loop (vmcnt(0) in every period)
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    prefetch_retire(pfs);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = b4[nt];
    __builtin_amdgcn_s_barrier();
    // ABL & 16: s_memtime stamps of workgroup 0 into p.ws: [wave][period][6] (scripts/wreg_stamps.py)
    unsigned long long* stamp = reinterpret_cast<unsigned long long*>(p.ws) + wave * 32 * 6;
    auto mark = [&](int q, int k) {
        if constexpr (ABL & 16) {
            if (bid == 0) {
                const unsigned long long t = __builtin_amdgcn_s_memtime();
                if (lane == 0) stamp[q * 6 + k] = t;
            }
        }
    };
    mark(0, 0);
    // period 0: tile 0 has no predecessor to activate
    mfmas(0, std::false_type{}, 3 < T && !(ABL & 4) ? 3 : -1);
    take();
    if (1 < T) wait_vmcnt_n(ndma(2) + ndma(3));
    __builtin_amdgcn_s_barrier();
    for (int q = 1; q < T; ++q) {
        mark(q, 0);
        mark(q, 1);
        // the 64 MFMAs of tile q with the activation of tile q-1 in their shadow: per MFMA two VALU instructions, a
        // ds_read_b128 every other one (two K steps ahead)
        mfmas(q, std::true_type{}, q + 3 < T && !(ABL & 4) ? q + 3 : -1);
        if constexpr (EPI != GP_EPI_GELU) activate();
        __builtin_amdgcn_sched_barrier(0);
        mark(q, 2);
        stores(q - 1);
        take();
        __builtin_amdgcn_sched_barrier(0);
        mark(q, 3);
        if (!ABL && q >= 3 && q + 3 < T) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");   // steady state: 2 x 4 DMA + 3 x 4 stores
        else if (q + 1 < T) wait_vmcnt_n(nst(q - 2) + ndma(q + 2) + nst(q - 1) + ndma(q + 3) + nst(q));
        mark(q, 4);
        __builtin_amdgcn_s_barrier();
        mark(q, 5);
    }
    activate();
    stores(T - 1);
}

// =====================================================================================================
// Weights-in-registers GEMM, second form (variant 17, round 4; the default for stage-2 fc1): the arithmetic of gemm_wreg_kernel
// (bitwise the same results, scripts/wreg2_ab.py and tests/test_hip_ops.py) with what its ablations and s_memtime stamps
// (profiles/r04_wreg_ablation.txt, r04_wreg_stamps.txt) pointed at:
//  * the stores were 8 bytes per lane (32-byte runs, four per tile and wave; the kernel without them: 75 of 85 us).  The 32
//    output channels of a wave are re-ordered -- row i of A fragment nt holds channel 8 (i / 4) + 4 nt + i % 4, a free choice of
//    which W rows a fragment is loaded from -- so that a lane's two n-tiles are 8 CONSECUTIVE channels: one 16-byte store per
//    lane and m-tile (64-byte runs), half the store instructions;
//  * one GELU slice of TWO chains behind every MFMA (24 of a half tile's 32 MFMAs) instead of four chains in lock step: 12
//    scratch registers instead of 24, no change in the instruction count;
//  * DMA lead 2 tiles (the pieces of tile t+2 ride in tile t): one whole tile for them to land, the wait in front of the barrier
//    never waits.
// Measured against variant 16 (interleaved medians, two boxes): 42.4 / 76.2 against 44.8 / 81.0 us at M = 16384 / 32768.  What
// was also built on this kernel and measured to be worth nothing or less (profiles/r04_wreg2_ab.txt, arms removed again): waves 4-7
// half a tile behind waves 0-3 with their barrier in the middle of the tile (+2-3 % TIME: MI355X_MICROARCH.md's stagger does
// not carry over to a tile whose every K step mixes MFMA and VALU), s_setprio 1 for waves 4-7 (nothing), B fragments 3 / 4
// instead of 2 K steps ahead (nothing: the LDS latency is not what it waits for), the four DMA pieces back to back behind the
// barrier (nothing).  The ablations say why: the MFMAs + fragment reads + barrier alone take 55 of the 85 us (1 250 TFLOP/s,
// the rate the best tile kernels of this library reach on random data), and GELU (10 us), stores (9) and DMA (9) ADD to that
// instead of hiding under it: on a SIMD with two waves of the same program, vector-issue time is the sum of both waves' MFMA,
// VALU, LDS and DMA instructions.
// LDS ring and hazards (NBUF = 4 tiles of 32 rows, tile t in buffer t & 3; b_t = the barrier that ends tile t):
//   RAW: every wave waits for ITS four DMA pieces of tile t+1 (counted vmcnt) in front of b_t.  WAR: buffer (t+2) & 3 held tile
//   t-2, whose last reads lie in front of b_t-2: the pieces of tile t+2 are issued during tile t.  vmcnt counts in issue order, DMA
//   pieces and stores alike; every store is ONE inline-asm instruction; allowed() counts what a wave issued behind the pieces it
//   waits for.
template <int EPI>
__global__ __launch_bounds__(512) void gemm_wreg2_kernel(const GemmKP p) {
    constexpr int K = 512, KS = K / 32, TM = 32, NBUF = 4, BUFB = TM * K * 2, NT = 2, MT = 2;
    __shared__ __attribute__((aligned(1024))) char smem[NBUF * BUFB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const int nsl = p.N >> 8;
    const int bid = blockIdx.x;
    const int item = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);   // XCD-contiguous chunks of the (m group, n slice) list (host: gridDim.x % 8 == 0)
    const int mg = item / nsl, ns = item - mg * nsl;
    const long m0 = (long)mg * p.tiles_m * TM;
    const int T = min(p.tiles_m, (int)((p.M - m0) / TM));
    if (T <= 0) return;
    const int nb = ns * 256 + wave * 32;

    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;
    const char* xsrc = reinterpret_cast<const char*>(p.X) + ((m0 + wave * 4) * (long)p.ldx) * 2;
    const long rowb = (long)p.ldx * 2;
    unsigned xoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xoff[i] = (unsigned)(i * rowb) + ((lane ^ ((wave * 4 + i) & 15)) << 4);
    auto piece = [&](int t, int i) {          // piece i (one row of 1 KB, chunk-swizzled on the source side) of this wave's four rows of tile t
        glds16_s(xsrc + (long)t * TM * rowb, xoff[i], lds0 + (t & (NBUF - 1)) * BUFB + (wave * 4 + i) * 1024);
    };
    // B fragment (mt, ks) of a tile: row mt*16 + fr, chunk (ks*4 + fq) ^ fr = ((ks & 3)*4 ^ (fq ^ fr)) + (ks >> 2) * 16
    unsigned boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) boff[j] = fr * 1024 + (((j * 4) ^ (fq ^ fr)) << 4);

    uint4 wf[NT][KS];
    f32x4 b4[NT], acc[NT][MT];
    f32x2 v[NT * MT * 2];   // pre-activation values of the previous tile: v[mt * 4 + nt * 2 + {0, 1}]
    // lane (fr, fq) owns row m0 + t TM + mt 16 + fr, channels nb + 8 fq .. + 8 (n-tile nt: + 4 nt .. + 4)
    half_t* Cw = reinterpret_cast<half_t*>(p.C) + (m0 + fr) * (long)p.ldc + nb + fq * 8;
    f32x2 gx[2], gt[2], gp[2];
    float c1v = GELU_H[1];
    asm volatile("" : "+v"(c1v));

    // in front of b_t a wave has issued, behind its pieces of tile t+1 (K steps 2..14 of tile t-1): stores(t-2), the pieces of tile
    // t+2 (if it exists) and stores(t-1)
    auto allowed = [&](int t) { return (t >= 2 ? MT : 0) + (t + 2 < T ? 4 : 0) + (t >= 1 ? MT : 0); };
    auto sync = [&](int t) {       // RAW wait for the own pieces of tile t+1, then the tile's barrier
        __builtin_amdgcn_sched_barrier(0);
        if (t >= 2 && t + 2 < T) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // steady state: 2 + 4 + 2
        else wait_vmcnt_n(allowed(t));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stores = [&](int t) {   // MT stores of 16 bytes per lane, one instruction each
        half_t* c = Cw + (long)t * TM * p.ldc;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x2 a = v[mt * 4], b = v[mt * 4 + 1], cc = v[mt * 4 + 2], d = v[mt * 4 + 3];
            const half8 o = {(half_t)a[0], (half_t)a[1], (half_t)b[0], (half_t)b[1], (half_t)cc[0], (half_t)cc[1], (half_t)d[0], (half_t)d[1]};
            const pf_u32x4 ov = __builtin_bit_cast(pf_u32x4, o);
            const half_t* dst = c + (long)mt * 16 * p.ldc;
            // (s_nop 1 inside the string: a store of more than 8 bytes reads its data registers for two more states, and hipcc pads
            // nothing around an asm statement -- its next v_cvt_pk into the same registers otherwise clobbers the fourth dword)
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(ov) : "memory");
        }
    };
    auto take = [&]() {   // accumulators -> v (v[mt * 4 + nt * 2 + h] = registers 2 h, 2 h + 1 of acc[nt][mt]), accumulators back to the bias
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const f32x4 a = acc[nt][mt];
                acc[nt][mt] = b4[nt];
                v[mt * 4 + nt * 2] = f32x2{a[0], a[1]};
                v[mt * 4 + nt * 2 + 1] = f32x2{a[2], a[3]};
            }
    };
    auto activate = [&]() {
        if constexpr (EPI == GP_EPI_GELU) {
            gelu_poly2_xn<4>(v);
            gelu_poly2_xn<4>(v + 4);
        } else if constexpr (EPI == GP_EPI_RELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = f32x2{fmaxf(v[i][0], 0.0f), fmaxf(v[i][1], 0.0f)};
        } else if constexpr (EPI == GP_EPI_LRELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = f32x2{v[i][0] > 0.0f ? v[i][0] : 0.1f * v[i][0], v[i][1] > 0.0f ? v[i][1] : 0.1f * v[i][1]};
        }
    };
    // one tile: 64 MFMAs with the activation of the previous tile's values in their shadow (K steps 0-7: v[0..3] = m-tile 0,
    // K steps 8-15: v[4..7] = m-tile 1; MFMA i of a half tile carries slice i % 12 of chain pair i / 12), B fragments read two
    // K steps ahead, one DMA piece of tile t+2 at K steps 2 / 6 / 10 / 14.  sched_barriers pin the order (hipcc otherwise clusters
    // the VALU work and waits lgkmcnt(0) in front of every MFMA pair).
    auto tile = [&](int t, auto shadow) {
        constexpr bool SH = decltype(shadow)::value && EPI == GP_EPI_GELU;
        const char* xb = smem + (t & (NBUF - 1)) * BUFB;
        uint4 bf[3][MT];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                bf[ks][mt] = *reinterpret_cast<const uint4*>(xb + boff[ks & 3] + mt * 16 * 1024 + (ks >> 2) * 256);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks + 2 < KS) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    bf[(ks + 2) % 3][mt] = *reinterpret_cast<const uint4*>(xb + boff[(ks + 2) & 3] + mt * 16 * 1024 + ((ks + 2) >> 2) * 256);
            }
            if constexpr (ks % 4 == 2) {
                if (t + 2 < T) piece(t + 2, ks / 4);
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, NT * MT>([&](auto jc) {
                constexpr int j = decltype(jc)::value, nt = j / MT, mt = j % MT;
                mma<half_t>(acc[nt][mt], wf[nt][ks], bf[ks % 3][mt]);
                if constexpr (SH) {
                    constexpr int half = ks / 8, i = (ks % 8) * 4 + j, cp = i / GELU_SLICES, slot = i % GELU_SLICES;
                    if constexpr (cp < 2) {
                        gelu_poly2_slice<slot>(v[half * 4 + cp * 2], gx[0], gt[0], gp[0], c1v);
                        gelu_poly2_slice<slot>(v[half * 4 + cp * 2 + 1], gx[1], gt[1], gp[1], c1v);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };

    // ---- prologue: tiles 0 and 1, the resident W fragments, the bias
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (t < T)
#pragma unroll
            for (int i = 0; i < 4; ++i) piece(t, i);
    const PfSink pfs = prefetch_issue(p, bid, gridDim.x, wave, 8, lane);
    // lane (fr, fq) of fragment (nt, ks) holds W[nb + 8 (fr / 4) + 4 nt + fr % 4][ks * 32 + fq * 8 .. + 8]
    const half_t* Wp = reinterpret_cast<const half_t*>(p.W) + (long)(nb + (fr >> 2) * 8 + (fr & 3)) * K + fq * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) wf[nt][ks] = *reinterpret_cast<const uint4*>(Wp + (long)nt * 4 * K + ks * 32);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        b4[nt] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + nb + fq * 8 + nt * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the builtin, so that hipcc does not wait for the W loads inside the loop
    prefetch_retire(pfs);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[nt][mt] = b4[nt];
    __builtin_amdgcn_s_barrier();         // tiles 0 and 1 are published

    clk_stamp(p, 0);
    tile(0, std::false_type{});
    take();
    sync(0);
    for (int t = 1; t < T; ++t) {
        tile(t, std::true_type{});
        if constexpr (EPI != GP_EPI_GELU) activate();
        __builtin_amdgcn_sched_barrier(0);
        stores(t - 1);
        take();
        sync(t);
    }
    clk_stamp(p, 1);
    activate();
    stores(T - 1);
}

// =====================================================================================================
// Weights-in-registers GEMM, third form (variant 19, round 5): gemm_wreg2_kernel on v_mfma_f32_32x32x16_f16.
// Why: the kernel is bound by the SIMD's vector ISSUE, not by its matrix pipe (header of gemm_wreg_kernel; docs/history/round4.md 9.1): an MFMA
// holds the issue port for 8 cycles whatever its shape, and a 32x32x16 MFMA does in 32 pipe cycles what two 16x16x32 MFMAs do in
// 2 x 16 -- half the MFMA issue slots per FLOP (per tile and wave 32 x 8 = 256 cycles instead of 64 x 8 = 512), the same fragment
// reads (one ds_read_b128 per K step of 16 instead of two per K step of 32) and the same registers (a wave's 32 W rows x 512 K are
// 32 A fragments = 128 VGPRs either way; 16 accumulators).  The GELU of the previous tile rides behind the MFMAs as before, six
// plain fp32 operations behind each (G16 = 0: the polynomial of gelu_poly2, same roundings), or -- G16 = 1 -- as 13 PACKED fp16
// operations per two values (gelu16_* below: v_pk_fma_f16 is not the packed-fp32 family of DESIGN.md 6b).
// Layouts: A fragment ks: lane l = W row perm(l % 32), k = 16 ks + 8 (l / 32) .. + 8; B fragment ks: pixel row l % 32 of the X tile,
// the same k.  Accumulator register r of lane l is A row 8 (r / 4) + 4 (l / 32) + r % 4, pixel l % 32; perm maps A row 8 g + 4 h + e
// to channel 16 (g / 2) + 8 h + 4 (g % 2) + e, so that a lane's registers 8 s .. 8 s + 7 are the 8 consecutive channels
// 16 s + 8 h .. + 8: two 16-byte stores per lane and tile, each instruction writing 32 rows x 32 contiguous bytes.
// LDS ring, hazards, vmcnt accounting: as gemm_wreg2_kernel (4 DMA pieces and 2 stores per tile and wave).
typedef float f32x16 __attribute__((ext_vector_type(16)));

// S32 = 1: v_mfma_f32_32x32x16_f16 (layouts in the header above); S32 = 0: v_mfma_f32_16x16x32_f16 with gemm_wreg2_kernel's layouts (the chip holds a
// ~10 % higher clock on that shape: scripts/kernel_clock.py).  Both: TWO accumulator sets -- tile t accumulates into set t & 1, its first
// MFMA per accumulator takes the bias registers as the C operand, and the activation of tile t - 1 reads set (t - 1) & 1 in place: the 32
// v_mov per tile and wave of gemm_wreg2_kernel's take() are gone (on this SIMD every VALU instruction adds ~4.5 issue cycles to the
// tile, two waves or one: scripts/probes/mfma_valu_coissue.hip).
template <int EPI, int S32, int G16>
__global__ __launch_bounds__(512) void gemm_wreg3_kernel(const GemmKP p) {
    constexpr int K = 512, KS = S32 ? K / 16 : K / 32, TM = 32, NBUF = 4, BUFB = TM * K * 2, NMF = S32 ? KS : 4 * KS;
    constexpr bool GELU = EPI == GP_EPI_GELU, H16 = GELU && G16;
    __shared__ __attribute__((aligned(1024))) char smem[NBUF * BUFB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5, fr = lane & 15, fq = lane >> 4;
    const int nsl = p.N >> 8;
    const int bid = blockIdx.x;
    const int item = (bid & 7) * (gridDim.x >> 3) + (bid >> 3);   // XCD-contiguous chunks of the (m group, n slice) list (host: gridDim.x % 8 == 0)
    const int mg = item / nsl, ns = item - mg * nsl;
    const long m0 = (long)mg * p.tiles_m * TM;
    const int T = min(p.tiles_m, (int)((p.M - m0) / TM));
    if (T <= 0) return;
    const int nb = ns * 256 + wave * 32;

    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;
    const char* xsrc = reinterpret_cast<const char*>(p.X) + ((m0 + wave * 4) * (long)p.ldx) * 2;
    const long rowb = (long)p.ldx * 2;
    unsigned xoff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xoff[i] = (unsigned)(i * rowb) + ((lane ^ ((wave * 4 + i) & 15)) << 4);
    auto piece = [&](int t, int i) {          // piece i (one row of 1 KB, chunk-swizzled on the source side) of this wave's four rows of tile t
        glds16_s(xsrc + (long)t * TM * rowb, xoff[i], lds0 + (t & (NBUF - 1)) * BUFB + (wave * 4 + i) * 1024);
    };
    // B fragments.  S32: K step ks (16 wide): row j, logical 16-byte chunk 2 ks + h, stored at chunk ^ (j & 15):
    //   byte j * 1024 + (ks / 8) * 256 + (((2 (ks % 8) + h) ^ (j & 15)) << 4).
    // 16x16x32: fragment (mt, ks): row mt * 16 + fr, chunk (4 ks + fq) ^ fr: byte fr * 1024 + mt * 16384 + (ks / 4) * 256 + (((4 (ks % 4)) ^ (fq ^ fr)) << 4)
    constexpr int NBO = S32 ? 8 : 4;
    unsigned boff[NBO];
#pragma unroll
    for (int q = 0; q < NBO; ++q) boff[q] = S32 ? j * 1024 + ((((2 * q) | h) ^ (j & 15)) << 4) : fr * 1024 + (((q * 4) ^ (fq ^ fr)) << 4);

    uint4 wf[S32 ? KS : 2 * KS];      // S32: [ks]; 16x16x32: [nt * KS + ks]
    f32x16 acc[2], b16;               // 16x16x32: registers 4 (nt * 2 + mt) .. + 4 of a set = accumulator (nt, mt); b16[4 nt .. + 4] = bias of n-tile nt
    unsigned o16[8];                  // G16: the packed fp16 results of the previous tile
    f32x2 v[8];                       // fp32 activations: the values of the previous tile, activated in place
    // value pair c of a set (two consecutive channels of one row) and where it goes:
    //   S32:      registers 2 c, 2 c + 1: row j, channels nb + 16 (c / 4) + 8 h + 2 (c % 4)       -> store s = c / 4, dword c % 4
    //   16x16x32: c = mt * 4 + nt * 2 + e: registers 4 (nt * 2 + mt) + 2 e, + 1: row mt * 16 + fr, channels nb + 8 fq + 4 nt + 2 e -> store mt, dword nt * 2 + e
    auto pair_reg = [](int c) constexpr { return S32 ? 2 * c : 4 * (((c >> 1) & 1) * 2 + (c >> 2)) + 2 * (c & 1); };
    half_t* Cw = reinterpret_cast<half_t*>(p.C) + (S32 ? (m0 + j) * (long)p.ldc + nb + h * 8 : (m0 + fr) * (long)p.ldc + nb + fq * 8);
    f32x2 gx[2], gt[2], gp[2];
    half2v hx[2], hu[2], hp[2];
    float c1v = GELU_H[1];
    asm volatile("" : "+v"(c1v));

    auto allowed = [&](int t) { return (t >= 2 ? 2 : 0) + (t + 2 < T ? 4 : 0) + (t >= 1 ? 2 : 0); };
    auto sync = [&](int t) {       // RAW wait for the own pieces of tile t+1, then the tile's barrier
        __builtin_amdgcn_sched_barrier(0);
        if (t >= 2 && t + 2 < T) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // steady state: 2 + 4 + 2
        else wait_vmcnt_n(allowed(t));
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    auto stores = [&](int t) {   // 2 stores of 16 bytes per lane, one instruction each
        half_t* c = Cw + (long)t * TM * p.ldc;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            pf_u32x4 ov;
            if constexpr (H16) {
                ov = pf_u32x4{o16[s2 * 4], o16[s2 * 4 + 1], o16[s2 * 4 + 2], o16[s2 * 4 + 3]};
            } else {
                const f32x2 a = v[s2 * 4], b = v[s2 * 4 + 1], cc = v[s2 * 4 + 2], d = v[s2 * 4 + 3];
                const half8 o = {(half_t)a[0], (half_t)a[1], (half_t)b[0], (half_t)b[1], (half_t)cc[0], (half_t)cc[1], (half_t)d[0], (half_t)d[1]};
                ov = __builtin_bit_cast(pf_u32x4, o);
            }
            const half_t* dst = S32 ? c + s2 * 16 : c + (long)s2 * 16 * p.ldc;
            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(dst), "v"(ov) : "memory");
        }
    };
    // activation of accumulator set `par` outside the MFMA shadow (the last tile; the cheap activations): -> v / o16
    auto activate = [&](auto parc) {
        constexpr int par = decltype(parc)::value;
        static_for<0, 8>([&](auto cc) {
            constexpr int c = decltype(cc)::value, r = pair_reg(c);
            v[c] = f32x2{acc[par][r], acc[par][r + 1]};
        });
        if constexpr (H16) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                half2v xh, u, pp;
                static_for<0, GELU16_SLICES>([&](auto sc) { gelu16_slice<decltype(sc)::value>(v[c], xh, u, pp); });
                o16[c] = __builtin_bit_cast(unsigned, pp);
            }
        } else if constexpr (GELU) {
            gelu_poly2_xn<4>(v);
            gelu_poly2_xn<4>(v + 4);
        } else if constexpr (EPI == GP_EPI_RELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = f32x2{fmaxf(v[i][0], 0.0f), fmaxf(v[i][1], 0.0f)};
        } else if constexpr (EPI == GP_EPI_LRELU) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = f32x2{v[i][0] > 0.0f ? v[i][0] : 0.1f * v[i][0], v[i][1] > 0.0f ? v[i][1] : 0.1f * v[i][1]};
        }
    };
    // one tile into accumulator set PAR, with the GELU of set PAR ^ 1 (the previous tile) in the shadow of its MFMAs: the 8 value pairs go
    // through the polynomial two chains at a time (call n of the tile = pair group n / (2 NSL), slice (n % (2 NSL)) / 2, chain n % 2), the
    // calls spread evenly over the NMF MFMAs (fp32: 96 calls of two operations; packed fp16: 104 calls of one).  B fragments are read two
    // K steps ahead, one DMA piece of tile t+2 in each quarter of the tile.
    auto tile = [&](int t, auto parc, auto shadow) {
        constexpr int PAR = decltype(parc)::value;
        constexpr bool SH = decltype(shadow)::value && GELU;
        constexpr int NSL = H16 ? GELU16_SLICES : GELU_SLICES, NCALL = 8 * NSL;
        const char* xb = smem + (t & (NBUF - 1)) * BUFB;
        auto shadow_calls = [&](auto ic) {      // the calls that ride behind MFMA i of the tile
            if constexpr (SH) {
                constexpr int i = decltype(ic)::value, n0 = i * NCALL / NMF, n1 = (i + 1) * NCALL / NMF;
                static_for<n0, n1>([&](auto nc) {
                    constexpr int n = decltype(nc)::value, g = n / (2 * NSL), m = n % (2 * NSL), slot = m / 2, ch = m % 2, c = 2 * g + ch, r = pair_reg(c);
                    if constexpr (H16) {
                        if constexpr (slot == 0) hx[ch] = half2v{(half_t)acc[PAR ^ 1][r], (half_t)acc[PAR ^ 1][r + 1]};
                        else gelu16_slice<slot>(v[0], hx[ch], hu[ch], hp[ch]);
                        if constexpr (slot == GELU16_SLICES - 1) o16[c] = __builtin_bit_cast(unsigned, hp[ch]);
                    } else {
                        if constexpr (slot == 0) v[c] = f32x2{acc[PAR ^ 1][r], acc[PAR ^ 1][r + 1]};
                        gelu_poly2_slice<slot>(v[c], gx[ch], gt[ch], gp[ch], c1v);
                    }
                });
            }
        };
        if constexpr (S32) {
            uint4 bf[3];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bf[ks] = *reinterpret_cast<const uint4*>(xb + boff[ks & 7] + (ks >> 3) * 256);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, KS>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                if constexpr (ks + 2 < KS) bf[(ks + 2) % 3] = *reinterpret_cast<const uint4*>(xb + boff[(ks + 2) & 7] + ((ks + 2) >> 3) * 256);
                if constexpr (ks % 8 == 4) {
                    if (t + 2 < T) piece(t + 2, ks / 8);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ks == 0) acc[PAR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(&wf[ks]), *reinterpret_cast<const half8*>(&bf[ks % 3]), b16, 0, 0, 0);
                else acc[PAR] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(&wf[ks]), *reinterpret_cast<const half8*>(&bf[ks % 3]), acc[PAR], 0, 0, 0);
                shadow_calls(std::integral_constant<int, ks>{});
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
            uint4 bf[3][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) bf[ks][mt] = *reinterpret_cast<const uint4*>(xb + boff[ks & 3] + mt * 16 * 1024 + (ks >> 2) * 256);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, KS>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value;
                if constexpr (ks + 2 < KS) {
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        bf[(ks + 2) % 3][mt] = *reinterpret_cast<const uint4*>(xb + boff[(ks + 2) & 3] + mt * 16 * 1024 + ((ks + 2) >> 2) * 256);
                }
                if constexpr (ks % 4 == 2) {
                    if (t + 2 < T) piece(t + 2, ks / 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<0, 4>([&](auto jc) {
                    constexpr int q = decltype(jc)::value, nt = q / 2, mt = q % 2, r0 = 4 * q;
                    f32x4 cin;
                    if constexpr (ks == 0) cin = f32x4{b16[4 * nt], b16[4 * nt + 1], b16[4 * nt + 2], b16[4 * nt + 3]};
                    else cin = f32x4{acc[PAR][r0], acc[PAR][r0 + 1], acc[PAR][r0 + 2], acc[PAR][r0 + 3]};
                    const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&wf[nt * KS + ks]), *reinterpret_cast<const half8*>(&bf[ks % 3][mt]), cin, 0, 0, 0);
                    acc[PAR][r0] = d[0]; acc[PAR][r0 + 1] = d[1]; acc[PAR][r0 + 2] = d[2]; acc[PAR][r0 + 3] = d[3];
                    shadow_calls(std::integral_constant<int, ks * 4 + q>{});
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
    };

    // ---- prologue: tiles 0 and 1, the resident W fragments, the bias
#pragma unroll
    for (int t = 0; t < 2; ++t)
        if (t < T)
#pragma unroll
            for (int i = 0; i < 4; ++i) piece(t, i);
    const PfSink pfs = prefetch_issue(p, bid, gridDim.x, wave, 8, lane);
    if constexpr (S32) {
        // lane (j, h) of fragment ks holds W[nb + perm(j)][ks * 16 + h * 8 .. + 8], perm(8 g + 4 hh + e) = 16 (g / 2) + 8 hh + 4 (g % 2) + e
        const int pj = 16 * (j >> 4) + 8 * ((j >> 2) & 1) + 4 * ((j >> 3) & 1) + (j & 3);
        const half_t* Wp = reinterpret_cast<const half_t*>(p.W) + (long)(nb + pj) * K + h * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[ks] = *reinterpret_cast<const uint4*>(Wp + ks * 16);
        // accumulator register r = channel 16 (r / 8) + 8 h + r % 8
#pragma unroll
        for (int r = 0; r < 16; ++r) b16[r] = p.bias ? p.bias[nb + 16 * (r >> 3) + 8 * h + (r & 7)] : 0.0f;
    } else {
        // lane (fr, fq) of fragment (nt, ks) holds W[nb + 8 (fr / 4) + 4 nt + fr % 4][ks * 32 + fq * 8 .. + 8]
        // (the bias FIRST: it is the C operand of tile 0's first MFMAs, and vmcnt retires in order -- behind the weights it would make them wait for the whole slice)
#pragma unroll
        for (int r = 0; r < 16; ++r) b16[r] = (p.bias && r < 8) ? p.bias[nb + fq * 8 + r] : 0.0f;   // registers 4 nt .. + 4: channels nb + 8 fq + 4 nt .. + 4
        asm volatile("" ::: "memory");
        const half_t* Wp = reinterpret_cast<const half_t*>(p.W) + (long)(nb + (fr >> 2) * 8 + (fr & 3)) * K + fq * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wf[nt * KS + ks] = *reinterpret_cast<const uint4*>(Wp + (long)nt * 4 * K + ks * 32);
    }
    // Round 6 (p.early): the first tile does NOT wait for the whole 256 KB weight slice.  The wave's eight X pieces of tiles 0 / 1 are its OLDEST vector-memory
    // operations (asm volatile with a memory clobber: nothing tracked moves in front of them) and the 32 weight fragments are ordinary loads hipcc tracks, issued in the
    // order tile 0 consumes them: vmcnt(32) says the pieces have landed whatever else is in flight, and hipcc's own waits in front of each fragment's first MFMA let tile 0
    // start on fragment 0 while fragments 1 .. 31 stream in.  (LDS-DMA instructions hipcc cannot see only make its counted waits stricter: extra YOUNGER operations.)
    // After tile 0 every fragment has been waited for, so no wait reaches the loop.  GP_GEMM_WREG_EARLY=0: the old full wait (A/B).
    if (p.early) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    else {
        __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0): the builtin, so that hipcc does not wait for the W loads inside the loop
        prefetch_retire(pfs);
    }
    __builtin_amdgcn_s_barrier();         // tiles 0 and 1 are published

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    clk_stamp(p, 0);
    tile(0, P0{}, std::false_type{});
    if (p.early) prefetch_retire(pfs);
    sync(0);
    int t = 1;
    for (; t + 1 < T; t += 2) {
        tile(t, P1{}, std::true_type{});
        if constexpr (!GELU) activate(P0{});
        __builtin_amdgcn_sched_barrier(0);
        stores(t - 1);
        sync(t);
        tile(t + 1, P0{}, std::true_type{});
        if constexpr (!GELU) activate(P1{});
        __builtin_amdgcn_sched_barrier(0);
        stores(t);
        sync(t + 1);
    }
    if (t < T) {          // T even: one more tile into set 1
        tile(t, P1{}, std::true_type{});
        if constexpr (!GELU) activate(P0{});
        __builtin_amdgcn_sched_barrier(0);
        stores(t - 1);
        sync(t);
        clk_stamp(p, 1);
        activate(P1{});
    } else {
        clk_stamp(p, 1);
        activate(P0{});
    }
    stores(T - 1);
}

template <typename T, int WM, int WN, int MT, int NT, int NS, bool DB = false, int RB = 128, bool PP = false, bool SPL = false, bool R32 = false> void launch_big(GemmKP& p, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * NT * 16;
    p.nkt = p.K / (RB / (int)sizeof(T));
    if constexpr (SPL) { p.split_n1 = p.nkt; p.nkt *= 3; }   // three K segments (GemmKP::split_n1)
    p.tiles_m = cdiv(p.M, BM);
    p.tiles_n = cdiv(p.N, BN);
    int splits = 1;
    if constexpr (!PP && !(DB && NS == 2)) {   // split-K: blockIdx.y walks the K ranges (generic ring schedule only)
        if (p.splitk > 1) {
            if (p.splitk > p.nkt) p.splitk = p.nkt;
            p.kt_per_split = cdiv(p.nkt, p.splitk);
            p.splitk = cdiv(p.nkt, p.kt_per_split);
            splits = p.splitk;
        }
    }
    hipLaunchKernelGGL((gemm_big_kernel<T, WM, WN, MT, NT, NS, DB, RB, PP, SPL, R32>), dim3(p.tiles_m * p.tiles_n, splits), dim3(WM * WN * 64), 0, s, p);
}


// ============================================================================ small-M GEMM (variant 18, round 4)
// The reference's real caller feeds PoseNet the detections of ONE frame (evaluation/evaluate.py:89-117): B = 1..8 crops, i.e.
// M = 256 B rows at stage 2 of the trunk (27 blocks) and 64 B at stage 3.  The tile kernels above are built for throughput: at M = 256
// a 128 x 128 tile leaves 32 workgroups on 256 CUs, each walking K through a global -> LDS -> register pipeline whose fill and
// drain ARE the kernel (10 / 15 us per fc1 / fc2 inside a hipGraph chain; split-K adds a reduce launch).  This kernel is built for
// latency instead:
//   * workgroup tile (16 MT) x 32, MT in {1, 2, 4} (chosen per launch by the cost model in gp_gemm): e.g. 32 x 32 for fc1 at one crop
//     (512 workgroups), 16 x 32 for fc2 (256);
//   * the four waves of a workgroup split K four ways; a wave loads its operand fragments STRAIGHT from global memory -- 16 bytes
//     per lane, row-major over the lanes (lane l: row l / 4, chunk l % 4 of the 32-wide K step: four lanes read 64 contiguous
//     bytes; X and W alike, both are K-contiguous) -- ALL of its K range at once (one wave per SIMD: up to 16 steps x 3
//     fragments x 4 = 192 registers in flight), so a wave pays ONE memory round trip, not one per pipeline stage; a ds_bpermute
//     per dword then puts each fragment into the MFMA operand layout (to_mfma_lanes below);
//   * the four partial tiles meet in LDS and are added in FIXED order (wave 0..3: deterministic, replay-bitwise, the same for
//     every MT), then bias / activation / gamma-residual exactly as the tile kernels' lean epilogue (epi_apply) and one 8-byte
//     store per lane; optionally the tile kernels' fused GroupNorm statistics (GN: 64-row tiles);
//   * the epilogue's bias / gamma / residual loads are issued before the K loop (their latency hides under the operand loads).
// Operand roles as everywhere in this file: W is the MFMA A operand and X the B operand, so a lane ends up with 4 consecutive
// n of one row m.  Convs (the heads' 3 x 3 256 -> 256 convs, the 2 x 2 down-sampling convs) run through the same kernel:
// a row is an output pixel, a 32-wide K step lies inside one filter tap (Cin % 32 == 0), out-of-image taps are masked to zero.

// a fragment loaded row-major over the lanes (lane l: row l / 4, chunk l % 4) -> the MFMA operand layout (lane l: row l % 16, chunk l / 16):
// MFMA lane (r, q) pulls the four dwords of lane 4 r + q through the LDS crossbar (no LDS memory involved)
__device__ __forceinline__ half8 to_mfma_lanes(half8 v, int pull) {
    pf_u32x4 u = __builtin_bit_cast(pf_u32x4, v);
#pragma unroll
    for (int d = 0; d < 4; ++d) u[d] = (unsigned)__builtin_amdgcn_ds_bpermute(pull, (int)u[d]);
    return __builtin_bit_cast(half8, u);
}

template <int MT, int NT, int KCH, bool CONV, bool GN>
__global__ __launch_bounds__(256) void gemm_smallm_kernel(const GemmKP p) {
    static_assert(!GN || NT == 2, "fused GroupNorm statistics: one workgroup = one (16 MT)-row chunk x 32 columns");
    extern __shared__ __attribute__((aligned(16))) char smallm_lds[];           // 4 waves x MT NT tiles x 64 lanes x 16 B (<= 32 KB)
    f32x4 (*red)[MT * NT][64] = reinterpret_cast<f32x4 (*)[MT * NT][64]>(smallm_lds);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;     // (wave stays a vector value: the conv's tap state in SGPRs spilled them)
    // LOADS: lane l fetches row l / 4, 16-byte chunk l % 4 of a K step -- four neighbouring lanes read 64 contiguous bytes, a quarter
    // wave 4 rows (in the MFMA layout, row l % 16 / chunk l / 16, a quarter wave touches sixteen 16-byte pieces of sixteen rows:
    // measured 15 B/clk/CU).  One ds_bpermute per dword then hands MFMA lane (r, q) the data lane 4 r + q loaded.
    const int rl = lane >> 2, ql = lane & 3, pull = (4 * r + q) * 4;
    const int n0 = blockIdx.x * (16 * NT), m0 = blockIdx.y * (16 * MT);
    const int nk = p.K >> 5;
    const int per = (nk + 3) >> 2, k_lo = wave * per, k_hi = min(nk, k_lo + per);
    const PfSink pf = prefetch_issue(p, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, wave, 4, lane);
    // ---- epilogue operands of the (tile, lane) pairs this thread finishes: wave w takes tiles w, w + 4, ...
    constexpr int TPW = (MT * NT + 3) / 4;
    f32x4 eb[TPW], eg[TPW];
    half4 er[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int et = wave + 4 * i;
        eb[i] = eg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        er[i] = half4{0, 0, 0, 0};
        if (et < MT * NT) {
            const int em = min(m0 + (et / NT) * 16 + r, p.M - 1), en = n0 + (et % NT) * 16 + q * 4;      // (plain GEMMs: M need not fill the last row tile)
            if (p.bias) eb[i] = *reinterpret_cast<const f32x4*>(p.bias + en);
            if (p.epi == GP_EPI_SCALE_RES) eg[i] = *reinterpret_cast<const f32x4*>(p.gamma + en);
            if (p.epi == GP_EPI_SCALE_RES || p.epi == GP_EPI_RES_RELU) er[i] = load_res4<half_t>(p, em, en);
        }
    }
    // ---- operand rows
    const half_t* wrow[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) wrow[nt] = reinterpret_cast<const half_t*>(p.W) + (long)(n0 + nt * 16 + rl) * p.K + ql * 8;
    const half_t* xrow[MT];
    int py[MT], px[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int m = m0 + mt * 16 + rl;
        if constexpr (CONV) {
            const int hw = p.Ho * p.Wo, b = m / hw, rem = m - b * hw;
            const int oy = rem / p.Wo;
            py[mt] = oy * p.stride - p.pad;                 // input row / column of filter tap (0, 0)
            px[mt] = (rem - oy * p.Wo) * p.stride - p.pad;
            xrow[mt] = reinterpret_cast<const half_t*>(p.X) + ((long)b * p.H * p.Win) * p.Cin + ql * 8;
        } else {
            py[mt] = px[mt] = 0;
            xrow[mt] = reinterpret_cast<const half_t*>(p.X) + (long)min(m, p.M - 1) * p.ldx + ql * 8;      // rows past M re-read the last row (never stored)
        }
    }
    const int cpt = CONV ? p.Cin >> 5 : 1;      // K steps per filter tap
    const int hm1 = p.H - 1, wm1 = p.Win - 1;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    // conv: (filter row, filter column, channel step) of the K step at hand, advanced step by step (no division per step)
    int dy = 0, dx = 0, cc = 0;
    if constexpr (CONV) {
        const int tap = k_lo / cpt;
        cc = k_lo - tap * cpt;
        dy = tap / p.KW;
        dx = tap - dy * p.KW;
    }
    for (int kb = k_lo; kb < k_hi; kb += KCH) {
        half8 wf[KCH][NT], xf[KCH][MT];
#pragma unroll
        for (int s = 0; s < KCH; ++s) {
            const int ks = min(kb + s, k_hi - 1);         // steps past the range reload the last one (never used)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) wf[s][nt] = *reinterpret_cast<const half8*>(wrow[nt] + (long)ks * 32);
            const int c0 = cc * 32;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                if constexpr (CONV) {
                    // (steps past the range: some tap past the last one -- in the image or masked, never used)
                    const int iy = py[mt] + dy, ix = px[mt] + dx;
                    // out-of-image taps: the load goes to pixel (0, 0) and is masked with a per-lane word built by integer arithmetic
                    // (sign bit of iy | H-1-iy | ix | W-1-ix): a compare + select keeps one 64-bit lane mask per fragment in
                    // SGPRs until the load lands -- 36 of them in the 64-row tile spilled SGPRs
                    const int keep = ~((iy | (hm1 - iy) | ix | (wm1 - ix)) >> 31);
                    const half_t* src = xrow[mt] + ((long)(iy & keep) * p.Win + (ix & keep)) * p.Cin + c0;
                    pf_u32x4 u = *reinterpret_cast<const pf_u32x4*>(src);
                    u &= pf_u32x4{(unsigned)keep, (unsigned)keep, (unsigned)keep, (unsigned)keep};
                    xf[s][mt] = __builtin_bit_cast(half8, u);
                } else {
                    xf[s][mt] = *reinterpret_cast<const half8*>(xrow[mt] + (long)ks * 32);
                }
            }
            if constexpr (CONV) {
                if (++cc == cpt) { cc = 0; if (++dx == p.KW) { dx = 0; ++dy; } }
            }
        }
        // every load of the round is issued before the first MFMA: left alone the scheduler interleaves them to save registers
        // (136 of the 192 the fragments need), which turns the one memory round trip the kernel is built around into several
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < KCH; ++s) {
            if (kb + s < k_hi) {                          // wave-uniform
                half8 wm[NT], xm[MT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) wm[nt] = to_mfma_lanes(wf[s][nt], pull);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) xm[mt] = to_mfma_lanes(xf[s][mt], pull);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wm[nt], xm[mt], acc[mt][nt], 0, 0, 0);
            }
        }
    }
    prefetch_retire(pf);                                 // (a wave with an empty K range waits for its prefetch loads here)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) red[wave][mt * NT + nt][lane] = acc[mt][nt];
    __syncthreads();
    float gs = 0.f, gq = 0.f;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int et = wave + 4 * i;
        if (et < MT * NT) {
            f32x4 v = red[0][et][lane];
            v += red[1][et][lane];
            v += red[2][et][lane];
            v += red[3][et][lane];
            v = epi_apply<half_t>(p.epi, v, eb[i], eg[i], er[i]);
            if constexpr (GN) {      // as the tile kernels: statistics of the fp32 values, before the fp16 rounding of the store
                gs += (v[0] + v[1]) + (v[2] + v[3]);
                gq += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
            }
            if (m0 + (et / NT) * 16 + r < p.M) store4<half_t>(p, m0 + (et / NT) * 16 + r, n0 + (et % NT) * 16 + q * 4, v);
        }
    }
    if constexpr (GN) {
        __shared__ float gred[4][4][2];
        // fused GroupNorm statistics: (sum, sum of squares) of this (16 MT)-row chunk per channel group, in the tile kernels' layout
        // (B, HW / (16 MT), G, 2) (gp_gemm_desc.gn_rows = 16 MT: 64 as the tile kernels, or -- round 5 -- 32 / 16 where few rows want more
        // workgroups and fewer load rounds).  Wave w finished the m-tiles w / 2 (+ 2) of n-tile w % 2 (MT = 1: waves 0 and 1 only): 16 rows by
        // shuffle, the group's second 4-column quad (8 channels per group) by shuffle, the waves of an n-tile through LDS in fixed order.
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { gs += __shfl_xor(gs, o, 64); gq += __shfl_xor(gq, o, 64); }
        if (p.gn_cpg == 8) { gs += __shfl_xor(gs, 16, 64); gq += __shfl_xor(gq, 16, 64); }
        if (r == 0) { gred[wave][q][0] = gs; gred[wave][q][1] = gq; }     // (a wave without a tile -- MT = 1: waves 2, 3 -- writes zeros, never read)
        __syncthreads();
        if (threadIdx.x < 8) {
            const int nt = threadIdx.x >> 2, qq = threadIdx.x & 3;
            if (p.gn_cpg == 4 || (qq & 1) == 0) {
                constexpr int ROWS = 16 * MT;
                const int G = p.N / p.gn_cpg, cpi = p.gn_hw / ROWS;
                const int b = m0 / p.gn_hw, ch = (m0 - b * p.gn_hw) / ROWS;
                float* o = p.gn_partial + (((long)b * cpi + ch) * G + (n0 + nt * 16 + qq * 4) / p.gn_cpg) * 2;
                if constexpr (MT == 1) {
                    o[0] = gred[nt][qq][0];
                    o[1] = gred[nt][qq][1];
                } else {
                    o[0] = gred[nt][qq][0] + gred[nt + 2][qq][0];
                    o[1] = gred[nt][qq][1] + gred[nt + 2][qq][1];
                }
            }
        }
    }
}

template <int MT, int NT, bool CONV, bool GN>
static void launch_smallm(const GemmKP& p, hipStream_t s) {
    // fragments in flight: (MT + NT) x KCH x 4 registers <= 256 (one wave per SIMD); convs: 9 (3 x 3 x Cin / 128 steps per wave come
    // in nines, and 16 unrolled steps of tap state spilled SGPRs); conv + fused GroupNorm statistics: 6 (9 spilled two SGPRs)
    // (MT <= 2 with fused statistics: 9, two rounds for the heads' 3 x 3 x 256 convs where the 64-row tile takes three of 6)
    constexpr int KMAX = (CONV && GN) ? (MT == 4 ? 6 : 9) : (CONV || MT + NT > 4) ? 9 : 16;
    const int per = (p.K / 32 + 3) / 4, rounds = (per + KMAX - 1) / KMAX, kch = (per + rounds - 1) / rounds;
    const dim3 grid(p.N / (16 * NT), (p.M + 16 * MT - 1) / (16 * MT));
    constexpr int LDS = 4 * MT * NT * 1024;
#define GP_SM(KCH)                                                                                                              \
    do {                                                                                                                        \
        auto kfn = gemm_smallm_kernel<MT, NT, (KCH) <= KMAX ? (KCH) : KMAX, CONV, GN>;                                          \
        if constexpr (LDS > 48 * 1024) {                                                                                        \
            static bool attr = false;                                                                                           \
            if (!attr) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; } \
        }                                                                                                                       \
        hipLaunchKernelGGL(kfn, grid, dim3(256), LDS, s, p);                                                                    \
    } while (0)
    if (kch <= 2) { GP_SM(2); return; }
    if (kch <= 4) { GP_SM(4); return; }
    if constexpr (!CONV) {          // (the 8-step conv instantiation of the 64-row tile spilled SGPRs: convs go 2 / 4 / 9, or 2 / 4 / 6)
        if (kch <= 8) { GP_SM(8); return; }
    }
    if constexpr (KMAX >= 9) {
        if (kch <= 9) { GP_SM(9); return; }
    }
    GP_SM(KMAX);
#undef GP_SM
}


// ============================================================================ row-vector GEMM (variant 23, round 5)
// M <= 8 rows -- ConvPnPNet's fc1 / fc2 over the detections of one frame (network/conv_pnp_net.py:186-199: M = the number of crops, fc1 is 2048 x 8192 = 32 MB of
// weights): the 128 x 128 tile kernel spends 14 + 8 us on it (split-K + reduce; one tile row of 128 for 1-8 used rows).  Here a wave owns TWO output columns and
// streams their weight rows straight into registers, 16 bytes per lane, every load of a 4 096-wide K block issued before anything waits; the few x rows go through
// LDS once per workgroup (8 columns); v_dot2_f32_f16 into fp32 accumulators per (row, column), a butterfly over the lanes, lane 0 applies bias / activation.
// One workgroup per 8 columns: 256 for fc1 = one per CU, each pulling 128 KB of weights.  Deterministic (fixed reduction order), fp16 in, fp16 or fp32 out.
template <int MR>
__global__ __launch_bounds__(256) void gemv_kernel(const GemmKP p) {
    constexpr int KC = 4096, NCH = KC / 512;
    __shared__ __attribute__((aligned(16))) half_t xs[MR][KC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 8 + wave * 2;
    const half_t* X = reinterpret_cast<const half_t*>(p.X);
    const half_t* W0 = reinterpret_cast<const half_t*>(p.W) + (long)n0 * p.K + lane * 8;
    float acc[MR][2];
#pragma unroll
    for (int m = 0; m < MR; ++m) acc[m][0] = acc[m][1] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += KC) {
        const int kc = min(KC, p.K - k0), nch = kc >> 9;          // K % 512 == 0 (host)
        uint4 w[2][NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            if (j < nch) {
                w[0][j] = *reinterpret_cast<const uint4*>(W0 + k0 + j * 512);
                w[1][j] = *reinterpret_cast<const uint4*>(W0 + p.K + k0 + j * 512);
            }
        if (k0) __syncthreads();
        for (int i = tid; i < MR * (kc >> 3); i += 256) {
            const int m = i / (kc >> 3), c = i - m * (kc >> 3);
            uint4 v = uint4{0u, 0u, 0u, 0u};
            if (m < p.M) v = *reinterpret_cast<const uint4*>(X + (long)m * p.ldx + k0 + c * 8);
            *reinterpret_cast<uint4*>(&xs[m][c * 8]) = v;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < NCH; ++j)
            if (j < nch) {
#pragma unroll
                for (int m = 0; m < MR; ++m) {
                    const uint4 xv = *reinterpret_cast<const uint4*>(&xs[m][j * 512 + lane * 8]);
                    const unsigned xw[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const unsigned ww[4] = {w[c][j].x, w[c][j].y, w[c][j].z, w[c][j].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[m][c] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, xw[e]), __builtin_bit_cast(half2v, ww[e]), acc[m][c], false);
                    }
                }
            }
    }
#pragma unroll
    for (int m = 0; m < MR; ++m)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) acc[m][c] += __shfl_xor(acc[m][c], o, 64);
    if (lane == 0) {
        const f32x4 z = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < MR; ++m)
            if (m < p.M) {
                const f32x4 b = p.bias ? f32x4{p.bias[n0], p.bias[n0 + 1], 0.f, 0.f} : z;
                const f32x4 v = epi_apply<half_t>(p.epi, f32x4{acc[m][0], acc[m][1], 0.f, 0.f}, b, z, half4{0, 0, 0, 0});
                if (p.out_f32) {
                    float* o = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n0;
                    o[0] = v[0];
                    o[1] = v[1];
                } else {
                    half_t* o = reinterpret_cast<half_t*>(p.C) + (long)m * p.ldc + n0;
                    o[0] = (half_t)v[0];
                    o[1] = (half_t)v[1];
                }
            }
    }
}

}  // namespace

// A/B switch for bench runs on one device: GP_GEMM_PP=0 keeps the ping-pong kernel out of the automatic choice
static bool pp_enabled() {
    static const bool on = [] { const char* e = getenv("GP_GEMM_PP"); return !(e && e[0] == '0'); }();
    return on;
}

static bool conv_wide_enabled() {   // A/B switch: GP_CONV_WIDE=0 keeps the 3x3 window conv on 256 x 256 tiles
    static const bool on = [] { const char* e = getenv("GP_CONV_WIDE"); return !(e && e[0] == '0'); }();
    return on;
}

static bool conv_window_enabled() {   // A/B switch: GP_CONV_WINDOW=0 keeps 3x3 convs on the tap-by-tap ping-pong kernel
    static const bool on = [] { const char* e = getenv("GP_CONV_WINDOW"); return !(e && e[0] == '0'); }();
    return on;
}

static long pp_min_k() {   // shortest K that goes to the ping-pong kernel when batches overlap (GP_PP_MIN_K, default 512);
                           // twice that for a launch that runs alone: measured end to end, bs 64 (K = 512: stage-2 fc1)
    static const long k = [] { const char* e = getenv("GP_PP_MIN_K"); return e ? atol(e) : 512l; }();
    return k;
}

static int gp_num_cus() {   // compute units of the current device (cached: 256 on MI355X)
    static const int n = [] {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
        return cu;
    }();
    return n;
}

static bool prefetch_enabled() {   // GP_GEMM_PREFETCH=0: A/B switch for gp_gemm_desc.prefetch
    static const bool on = [] { const char* e = getenv("GP_GEMM_PREFETCH"); return !(e && e[0] == '0'); }();
    return on;
}

static bool wreg_enabled() {
    static const bool on = [] { const char* e = getenv("GP_GEMM_WREG"); return !(e && e[0] == '0'); }();
    return on;
}
static bool wreg_early() {
    static const bool on = [] { const char* e = getenv("GP_GEMM_WREG_EARLY"); return !(e && e[0] == '0'); }();
    return on;
}
static long wreg_min_rows() {
    static const long k = [] { const char* e = getenv("GP_GEMM_WREG_MIN_ROWS"); return e ? atol(e) : 6144l; }();
    return k;
}

static int wreg_variant() {   // GP_GEMM_WREG_VARIANT=17|19|20|21|22: which weights-in-registers kernel the automatic choice takes (A/B switch; default 21,
                              // and 22 -- the same kernel with the activation in fp32 -- for the epilogues other than GELU)
    static const int v = [] { const char* e = getenv("GP_GEMM_WREG_VARIANT"); const int x = e ? atoi(e) : (gp_gelu16_enabled() ? 21 : 22); return (x >= 19 && x <= 22) ? x : 17; }();
    return v;
}

static long co_min_tiles() {   // fewest 256 x 256 tiles that still go to the ping-pong kernel when batches overlap (GP_GEMM_CO_MIN_TILES: A/B)
    static const long k = [] { const char* e = getenv("GP_GEMM_CO_MIN_TILES"); return e ? atol(e) : 32l; }();
    return k;
}

static double smallm_tile_bias() {   // GP_GEMM_SMALLM_BIAS=<us>: added to the tile kernels' estimate in the choice above (A/B switch)
    static const double b = [] { const char* e = getenv("GP_GEMM_SMALLM_BIAS"); return e ? atof(e) : 0.0; }();
    return b;
}
// the latency kernel's cost model (us), see gp_gemm: tile (16 mt) x 32
static double smallm_estimate(int M, int N, int K, int mt) {
    const double wgs = (double)(M / (16 * mt)) * (N / 32), kb_per_wg = (16.0 * mt + 32.0) * K * 2.0 / 1024.0;
    const double one = 1.0 + kb_per_wg / 60.0, many = wgs / 256.0 * kb_per_wg / 45.0;
    return 2.6 + (one > many ? one : many) + (mt == 4 ? 1.5 : 0.0);
}

static bool gemv_enabled() {   // GP_GEMM_GEMV=0: A/B switch, keeps the row-vector kernel (variant 23) out of the automatic choice
    static const bool on = [] { const char* e = getenv("GP_GEMM_GEMV"); return !(e && e[0] == '0'); }();
    return on;
}

static int smallm_max_rows() {   // rows up to which variant 0 may pick the latency kernel; GP_GEMM_SMALLM_MAXROWS=<n>: A/B switch
    static const int n = [] { const char* e = getenv("GP_GEMM_SMALLM_MAXROWS"); return e && atoi(e) > 0 ? atoi(e) : 16384; }();
    return n;
}

static bool smallm_enabled() {   // GP_GEMM_SMALLM=0: A/B switch, keeps the latency kernel (variant 18) out of the automatic choice
    static const bool on = [] { const char* e = getenv("GP_GEMM_SMALLM"); return !(e && e[0] == '0'); }();
    return on;
}

static long pp_min_tiles() {
    const char* e = getenv("GP_GEMM_PP_MIN_TILES");
    return e ? atol(e) : 192;
}

// Rows per fused-GroupNorm statistics chunk for an fp16 GEMM / conv of this shape (gp_gemm_desc.gn_rows; the consumers take chunks = HW / rows): 64 -- the
// tile kernels' chunk -- unless the latency kernel would take the launch and a smaller tile is faster by its cost model (few rows: the detections of one
// frame; 16-row tiles give the heads' 3 x 3 convs at one crop 128 workgroups and two load rounds instead of 32 and three).
extern "C" int gp_gemm_gn_rows(int M, int N, int K, int hw) {
    if (M <= 0 || N <= 0 || K <= 0 || N % 32 || M % 16 || hw % 64 || !smallm_enabled() || M > smallm_max_rows()) return 64;
    double best = 1e30;
    int mt_best = 4;
    for (int mt = 1; mt <= 4; mt *= 2) {
        if (M % (16 * mt)) continue;
        const double t = smallm_estimate(M, N, K, mt);
        if (t < best) { best = t; mt_best = mt; }
    }
    if (best > 30.0 + smallm_tile_bias()) return 64;
    return 16 * mt_best;
}

extern "C" int gp_gemm(const gp_gemm_desc* d, void* stream) {
    GP_REQUIRE(d != nullptr, "gp_gemm: null descriptor");
    GP_REQUIRE(d->dtype == GP_F32 || d->dtype == GP_F16, "gp_gemm: bad dtype %d", d->dtype);
    const int esz = d->dtype == GP_F16 ? 2 : 4;
    const int KPT = 128 / esz;
    GP_REQUIRE(d->X && d->W && d->C, "gp_gemm: null operand");
    GP_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gp_gemm: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
    GP_REQUIRE(d->N % 4 == 0, "gp_gemm: N=%d must be a multiple of 4", d->N);
    GP_REQUIRE(d->K % KPT == 0, "gp_gemm: K=%d must be a multiple of %d", d->K, KPT);
    GP_REQUIRE(d->ldc % 4 == 0 && d->ldc >= d->N, "gp_gemm: ldc=%d invalid", d->ldc);
    GP_REQUIRE(d->epilogue >= GP_EPI_NONE && d->epilogue <= GP_EPI_LNFOLD_GELU, "gp_gemm: bad epilogue");
    if (d->epilogue == GP_EPI_LNFOLD_GELU) {   // lean epilogue of the large-tile kernels only (every tile interior)
        GP_REQUIRE(d->dtype == GP_F16 && !d->out_f32 && d->splitk <= 1 && !d->gn_partial && d->KH == 0,
                   "gp_gemm: LNFOLD_GELU needs a plain fp16 GEMM without split-K / fused GroupNorm");
        GP_REQUIRE(d->bias && d->ln_stats && d->ln_colsum && d->ln_nslab > 0, "gp_gemm: LNFOLD_GELU needs bias, ln_stats, ln_colsum");
        GP_REQUIRE(d->M % 256 == 0 && d->N % 256 == 0 && d->ldc % 8 == 0 && ((size_t)d->C & 15) == 0 &&
                   ((size_t)d->ln_stats & 15) == 0 && ((size_t)d->ln_colsum & 15) == 0 && ((size_t)d->bias & 15) == 0,
                   "gp_gemm: LNFOLD_GELU needs M %% 256 == 0, N %% 256 == 0 and 16-byte aligned operands");
        GP_REQUIRE(d->variant % 100 == 0 || d->variant % 100 == 8 || d->variant % 100 == 10 || d->variant % 100 == 12,
                   "gp_gemm: LNFOLD_GELU runs on variants 8 / 10 / 12");
    }
    if (d->epilogue == GP_EPI_SCALE_RES) GP_REQUIRE(d->gamma != nullptr, "gp_gemm: SCALE_RES needs gamma");
    if (d->epilogue == GP_EPI_SCALE_RES || d->epilogue == GP_EPI_RES_RELU)
        GP_REQUIRE(d->residual && d->ldres % 4 == 0 && d->ldres >= d->N, "gp_gemm: epilogue needs a residual");
    GemmKP p;
    memset(&p, 0, sizeof(p));
    p.X = d->X; p.W = d->W; p.bias = d->bias; p.gamma = d->gamma; p.res = d->residual; p.C = d->C; p.ws = d->workspace;
    p.M = d->M; p.N = d->N; p.K = d->K; p.ldx = d->ldx; p.ldc = d->ldc; p.ldres = d->ldres;
    p.epi = d->epilogue; p.out_f32 = d->out_f32;
    p.ln_stats = d->ln_stats; p.ln_s = d->ln_colsum; p.ln_nsl = d->ln_nslab; p.ln_eps = d->ln_eps;
    if (d->prefetch && d->prefetch_bytes > 0 && prefetch_enabled()) { p.pf = reinterpret_cast<const char*>(d->prefetch); p.pf_bytes = d->prefetch_bytes; }
    const bool r32 = d->residual_f32 != 0;       // fp16 operands, fp32 residual (and output): the fp32 residual stream of the fp16 mode
    if (r32) GP_REQUIRE(d->dtype == GP_F16 && d->out_f32 && d->split_shift == 0 && d->splitk <= 1 && d->residual,
                        "gp_gemm: residual_f32 needs dtype GP_F16, out_f32, a residual, no split-K");
    if (d->c16) {
        GP_REQUIRE(d->out_f32 && d->dtype == GP_F16 && d->splitk <= 1 && d->ldc16 >= d->N && d->ldc16 % 4 == 0 && !d->out_planes,
                   "gp_gemm: c16 (fp16 copy of an fp32 output) needs dtype GP_F16, out_f32, ldc16 >= N, no split-K");
        p.c16 = d->c16; p.ldc16 = d->ldc16;
    }
    const bool split = d->split_shift > 0;
    if (split) {   // split-operand mode: fp16 hi / lo' planes in, fp32 out (GemmKP::split_n1)
        GP_REQUIRE(d->dtype == GP_F16 && d->out_f32, "gp_gemm: split-operand mode takes fp16 planes (dtype GP_F16) and writes fp32 (out_f32)");
        GP_REQUIRE(d->split_shift <= 14 && d->x_plane_stride > 0 && d->w_plane_stride >= (long)d->N * d->K,
                   "gp_gemm: split-operand mode needs split_shift <= 14 and the plane strides");
        GP_REQUIRE(d->epilogue != GP_EPI_LNFOLD_GELU, "gp_gemm: split-operand mode has no LNFOLD epilogue");
        GP_REQUIRE(d->x_plane_stride % 8 == 0 && d->w_plane_stride % 8 == 0, "gp_gemm: plane strides must keep 16-byte alignment");
        p.xplane_b = d->x_plane_stride * 2; p.wplane_b = d->w_plane_stride * 2;
        p.split_scale = ldexpf(1.0f, -d->split_shift);
        if (d->out_planes) {
            GP_REQUIRE(d->c_plane_stride >= (long)d->M * d->ldc && d->c_plane_stride % 4 == 0 && d->splitk <= 1,
                       "gp_gemm: out_planes needs c_plane_stride >= M * ldc (elements) and no split-K");
            p.out_planes = 1; p.cplane = d->c_plane_stride; p.split_up = ldexpf(1.0f, d->split_shift);
        }
    } else {
        GP_REQUIRE(!d->out_planes, "gp_gemm: out_planes belongs to the split-operand mode");
    }
    if (d->gn_partial) {
        GP_REQUIRE(d->gn_groups > 0 && d->N % d->gn_groups == 0 && (d->N / d->gn_groups == 4 || d->N / d->gn_groups == 8),
                   "gp_gemm: fused GroupNorm needs 4 or 8 channels per group");
        GP_REQUIRE(d->gn_hw > 0 && d->gn_hw % 64 == 0 && d->M % d->gn_hw == 0, "gp_gemm: fused GroupNorm needs HW %% 64 == 0");
        GP_REQUIRE(d->gn_rows == 0 || d->gn_rows == 16 || d->gn_rows == 32 || d->gn_rows == 64, "gp_gemm: gn_rows is 0 (= 64), 16, 32 or 64");
        GP_REQUIRE(d->splitk <= 1, "gp_gemm: fused GroupNorm excludes split-K");
        p.gn_partial = d->gn_partial; p.gn_cpg = d->N / d->gn_groups; p.gn_hw = d->gn_hw;
    }
    if (d->KH > 0) {
        GP_REQUIRE(d->KW > 0 && d->stride > 0 && d->pad >= 0, "gp_gemm: bad conv geometry");
        GP_REQUIRE(d->Cin % KPT == 0, "gp_gemm: conv Cin=%d must be a multiple of %d", d->Cin, KPT);
        GP_REQUIRE(d->K == d->KH * d->KW * d->Cin, "gp_gemm: conv K=%d != KH*KW*Cin", d->K);
        GP_REQUIRE(d->KH * d->KW <= 32, "gp_gemm: at most 32 filter taps");
        const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->Win + 2 * d->pad - d->KW) / d->stride + 1;
        GP_REQUIRE(Ho == d->Ho && Wo == d->Wo, "gp_gemm: conv output %dx%d != expected %dx%d", d->Ho, d->Wo, Ho, Wo);
        GP_REQUIRE((long)d->B * Ho * Wo == d->M, "gp_gemm: conv M=%d != B*Ho*Wo", d->M);
        p.conv = 1; p.H = d->H; p.Win = d->Win; p.Cin = d->Cin; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
        p.Ho = Ho; p.Wo = Wo;
    } else {
        GP_REQUIRE(d->ldx >= d->K && d->ldx % (16 / esz) == 0, "gp_gemm: ldx=%d invalid", d->ldx);
    }
    p.splitk = d->splitk > 1 ? d->splitk : 1;
    if (p.splitk > d->K / KPT * (split ? 3 : 1)) p.splitk = d->K / KPT * (split ? 3 : 1);
    if (p.splitk > 1) GP_REQUIRE(d->workspace != nullptr, "gp_gemm: splitk needs a workspace");
    hipStream_t s = (hipStream_t)stream;
    const double flops = 2.0 * d->M * d->N * d->K;
    const double xbytes = d->KH > 0 ? (double)d->B * d->H * d->Win * d->Cin * esz : (double)d->M * d->K * esz;
    const double bytes = xbytes + (double)d->N * d->K * esz + (double)d->M * d->N * (d->out_f32 ? 4 : esz) +
                         (d->epilogue >= GP_EPI_SCALE_RES ? (double)d->M * d->N * esz : 0.0);
    gp_timing_before(s, GP_KC_GEMM, flops, bytes);
    // K = 512 with the weights resident in registers (variant 16)
    const bool wreg_ok = d->dtype == GP_F16 && d->KH == 0 && d->K == 512 && d->N % 256 == 0 && d->M % 32 == 0 && !d->out_f32 &&
                         p.splitk == 1 && !d->gn_partial && d->epilogue <= GP_EPI_LRELU && d->ldc % 4 == 0 &&
                         ((size_t)d->C & 7) == 0 && ((size_t)d->X & 15) == 0 && ((size_t)d->W & 15) == 0 &&
                         (!d->bias || ((size_t)d->bias & 15) == 0);
    // few rows (the detections of one frame): the latency kernel (variant 18), whatever split-K factor the caller worked out for the
    // tile kernels.  GP_GEMM_SMALLM=0 keeps it out of the automatic choice (A/B switch)
    const bool smallm_ok = d->dtype == GP_F16 && !split && !r32 && !d->c16 && d->epilogue != GP_EPI_LNFOLD_GELU &&
                           (!d->out_f32 || (d->epilogue <= GP_EPI_LRELU && !d->gn_partial && ((size_t)d->C & 15) == 0)) &&      // fp32 out: the activation epilogues (round 5)
                           (!d->gn_partial || (d->M % (d->gn_rows ? d->gn_rows : 64) == 0 && d->epilogue != GP_EPI_SCALE_RES && d->epilogue != GP_EPI_RES_RELU)) &&
                           d->N % 32 == 0 && (d->M % 16 == 0 || (d->KH == 0 && !d->gn_partial && d->M > 8)) && (d->KH == 0 || d->Cin % 32 == 0) && ((size_t)d->X & 15) == 0 &&      // (plain GEMMs: any M > 8 -- the PnP fc layers have M = the crop count; round 6)
                           ((size_t)d->W & 15) == 0 && ((size_t)d->C & 7) == 0 && (!d->bias || ((size_t)d->bias & 15) == 0) &&
                           (!d->gamma || ((size_t)d->gamma & 15) == 0) && (!d->residual || ((size_t)d->residual & 7) == 0);
    // variant: 4 = 128x128 LDS-DMA (+split-K), 2 = 256x128, 3 = 256x256, 7..13 see below, 0 = pick
    int variant = d->variant % 100;
    p.dbg = d->variant / 100;
    p.early = wreg_early() ? 1 : 0;
    // Workgroup tile of the latency kernel: (16 MT) x 32, MT in {1, 2, 4}, by a cost model fitted to scripts/small_m_variants.py
    // (profiles/r04_small_m_tiles.txt; hipGraph chains, 1-8 crops, K 512-4096): ~2.6 us of launch + epilogue; ONE workgroup moves
    // its (16 MT + 32) rows of K halfs in ~1 us + KB / 60; a CU that holds several workgroups sustains ~45 KB/us; the 64-row
    // tile pays ~1.5 us more (two output tiles per wave, 32 KB through LDS).  The tile kernels (with the caller's split-K)
    // cost ~9 us + K / 160, never more than ~22 -- ~30 with fused GroupNorm statistics (their 128 x 128 tile + statistics epilogue is
    // 32 us flat up to 4 096 rows): the latency kernel takes the launch when its estimate is below that.  Fused GroupNorm
    // statistics need the 64-row tile.
    // M <= 8 rows of a plain GEMM: the row-vector kernel (variant 23; GP_GEMM_GEMV=0 keeps it out of the automatic choice)
    const bool gemv_ok = d->dtype == GP_F16 && !split && !r32 && !d->c16 && d->KH == 0 && !d->gn_partial && d->M <= 8 && d->K % 512 == 0 && d->N % 8 == 0 &&
                         d->ldx % 8 == 0 && d->epilogue <= GP_EPI_LRELU && ((size_t)d->X & 15) == 0 && ((size_t)d->W & 15) == 0 && ((size_t)d->C & 3) == 0 && d->ldc % 2 == 0;
    if (variant == 23 || (variant == 0 && gemv_ok && gemv_enabled())) {
        GP_REQUIRE(gemv_ok, "gp_gemm: variant 23 needs a plain fp16 GEMM of M <= 8 rows, K %% 512 == 0, N %% 8 == 0, bias / GELU / ReLU / LeakyReLU epilogue");
        p.splitk = 1;
        gp_timing_label("gemm v23 M%d N%d K%d epi%d", d->M, d->N, d->K, d->epilogue);
        const dim3 grid(d->N / 8);
        if (d->M <= 1) hipLaunchKernelGGL(gemv_kernel<1>, grid, dim3(256), 0, s, p);
        else if (d->M <= 2) hipLaunchKernelGGL(gemv_kernel<2>, grid, dim3(256), 0, s, p);
        else if (d->M <= 4) hipLaunchKernelGGL(gemv_kernel<4>, grid, dim3(256), 0, s, p);
        else hipLaunchKernelGGL(gemv_kernel<8>, grid, dim3(256), 0, s, p);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    int sm_mt = 0;
    // (smallm_max_rows(): 16 384 since the end of round 5 -- at 32 768 rows the estimate was 21 us for a launch that takes 32 where a tile kernel takes 16;
    // at 16 384 rows the latency kernel still wins, 11.6 against 15.6 us: profiles/r05_smallm_cap_ab.txt.  Before:)
    // (the estimate below was fitted on 1-8 crops: the automatic choice stops at 32 768 rows -- 16 crops' worth of the widest map it was
    // measured on -- whatever the estimate says beyond; tests/test_hip_posenet.py pins which launches take it at 4 / 8 / 16 crops)
    if (smallm_ok && smallm_enabled() && d->M <= smallm_max_rows()) {
        double best = 1e30;
        const int gn_mt = d->gn_partial ? (d->gn_rows ? d->gn_rows / 16 : 4) : 0;      // fused statistics: the caller's chunk rows ARE the tile rows
        for (int mt = gn_mt ? gn_mt : 1; mt <= (gn_mt ? gn_mt : 4); mt *= 2) {
            if (d->M % (16 * mt) && (d->M % 16 == 0 || mt > 2)) continue;      // (a plain GEMM whose M is no multiple of 16 runs 16- / 32-row tiles with a partial last one)
            const double t = smallm_estimate((d->M + 16 * mt - 1) / (16 * mt) * (16 * mt), d->N, d->K, mt);
            if (t < best) { best = t; sm_mt = mt; }
        }
        const double tile = (d->gn_partial ? 30.0 : 9.0 + d->K / 160.0 < 22.0 ? 9.0 + d->K / 160.0 : 22.0) + smallm_tile_bias();
        if (best > tile && !(gn_mt && gn_mt < 4)) sm_mt = 0;      // (statistics chunks of 16 / 32 rows exist in this kernel only: gp_gemm_gn_rows made the choice)
    }
    if ((variant == 0 && sm_mt) || variant == 18) {
        variant = 18;
        p.splitk = 1;
        if (!sm_mt) sm_mt = d->M % 64 == 0 && (d->M >= 2048 || d->gn_partial) ? 4 : d->M % 32 == 0 && d->M > 512 ? 2 : 1;     // explicit request: by row count
        if (d->gn_partial && d->gn_rows) sm_mt = d->gn_rows / 16;
        // variants 218 / 318 / 418: the tile forced to 16 x 32 / 32 x 32 / 64 x 32 (tests, scripts/small_m_variants.py)
        if (p.dbg >= 2 && p.dbg <= 4 && !d->gn_partial && d->M % (8 << (p.dbg - 1)) == 0) sm_mt = 1 << (p.dbg - 2);
    }
    if (variant == 0) {
        // measured per shape (scripts/gemm_bench.py): where 256x256 tiles fill the chip (N % 256 == 0, >= 192 tiles)
        // the ping-pong kernel (10) for K >= 512 and the 256x128 tile at two workgroups per CU (8) for shorter K
        // (its epilogues overlap the other workgroup's main loop); otherwise
        // 128x128 at two workgroups per CU (7); fp32 storage: 128x128 (4); split-K: 128x128 (4) + reduce kernel
        const long tA = (long)cdiv(d->M, 256) * cdiv(d->N, 256);
        if (p.splitk > 1) variant = 4;
        else if (d->dtype == GP_F16) {
            // co_scheduled (several batches in flight: PoseNet(inflight > 1)): a launch need not fill the chip by itself,
            // the 256x256 ping-pong tile is then the cheapest per FLOP even for 32-128 tiles (GP_GEMM_PP_MIN_TILES: A/B)
            const bool fills = tA >= 192 || d->epilogue == GP_EPI_LNFOLD_GELU;
            // K = 512, >= 6144 rows, >= 1024 columns (stage-2 fc1 from 24 crops up): weights in registers (bs 64: 48.5 us against 51.4 for
            // the ping-pong tile and 59-61 for the 256x128 tile, scripts/wreg_bench.py; round 6, scripts/midsize_variants.py: 24 crops 22.8 against 25.4 us
            // for the 256x128 tile, 32 crops 26.8 against 28.9; at 16 crops the 128x128 tile wins, 16.6 against 19.0: shorter M does not amortise the
            // 256 KB weight prologue per CU).  GP_GEMM_WREG=0 / GP_GEMM_WREG_MIN_ROWS=<n>: A/B switches
            // (split-operand mode: the K loop is three times as long, which is what the ping-pong tile's fill / drain is weighed
            // against; measured, scripts/split_variants.py: it wins from ~140 tiles of 256 x 256 up, below that the 128 x 128
            // two-workgroups-per-CU tile does; the 256 x 128 tile never)
            if (split) variant = (d->N % 256 == 0 && tA >= 140 && pp_enabled()) ? 10 : 7;
            else if (wreg_ok && d->M >= wreg_min_rows() && d->N >= 1024 && wreg_enabled()) variant = (d->ldc % 8 == 0 && ((size_t)d->C & 15) == 0) ? (wreg_variant() == 20 && d->epilogue != GP_EPI_GELU ? 19 : wreg_variant() == 21 && d->epilogue != GP_EPI_GELU ? 22 : wreg_variant()) : 16;
            else if (d->N % 256 == 0 && d->K >= (d->co_scheduled ? pp_min_k() : 2 * pp_min_k()) && pp_enabled() && (fills || (d->co_scheduled && tA >= co_min_tiles()) || tA >= pp_min_tiles())) variant = 10;
            else variant = (d->N % 256 == 0 && fills) ? 8 : 7;
        }
        else variant = 4;
    }
    if (r32 && d->variant % 100 == 0 && variant != 10) variant = 7;
    // 3x3 s1 p1, Cout 256, whole image rows per 256-pixel tile: the LDS-window kernel (variant 13; automatic when the
    // ping-pong kernel would have been chosen)
    const bool win_ok = d->dtype == GP_F16 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->N == 256 &&
                        d->Cin % 32 == 0 && (d->Win == 64 || d->Win == 32 || d->Win == 16) && d->H % (256 / d->Win) == 0 &&
                        (!d->out_f32 || split) && p.splitk == 1 && d->ldc % 8 == 0 && ((size_t)d->C & 15) == 0 && !d->out_planes &&
                        (d->epilogue == GP_EPI_NONE || d->epilogue == GP_EPI_GELU || d->epilogue == GP_EPI_RELU);
    if (variant == 10 && d->variant % 100 == 0 && win_ok && conv_window_enabled()) variant = 13;
    GP_REQUIRE(!(d->gn_partial && d->gn_rows && d->gn_rows != 64) || variant == 18, "gp_gemm: gn_rows=%d needs the small-M kernel (variant 18: fp16, N %% 32 == 0, M %% gn_rows == 0), got variant %d", d->gn_rows, variant);
    GP_REQUIRE(((variant >= 2 && variant <= 13 && variant != 6) || (variant >= 16 && variant <= 22)) && (variant == 4 || p.splitk == 1), "gp_gemm: bad variant %d (split-K runs on variant 4)", variant);
    GP_REQUIRE(!split || variant == 4 || variant == 7 || variant == 8 || variant == 10 || variant == 13, "gp_gemm: split-operand mode runs on variants 4 / 7 / 8 / 10 / 13 (got %d)", variant);
    GP_REQUIRE(!r32 || variant == 7 || variant == 10, "gp_gemm: residual_f32 runs on variants 7 / 10 (got %d)", variant);
    // c16 is written by the generic epilogue of gemm_big_kernel only (out_f32 keeps every tile off the lean one); the window conv
    // and the weights-in-registers kernel would return without it
    GP_REQUIRE(!d->c16 || (variant != 13 && variant != 16 && variant != 17 && !split), "gp_gemm: c16 runs on the tile kernels (variants 2-5, 7-12), not on %d%s", variant, split ? " split" : "");
    if (d->KH > 0) gp_timing_label("conv%dx%d s%d v%d %dx%d Cin%d Cout%d M%d%s%s", d->KH, d->KW, d->stride, variant, d->H, d->Win, d->Cin, d->N, d->M, d->gn_partial ? (d->gn_rows == 16 ? " +gn16" : d->gn_rows == 32 ? " +gn32" : " +gn") : "", split ? " split3" : "");
    else gp_timing_label("gemm v%d M%d N%d K%d epi%d%s%s%s", variant, d->M, d->N, d->K, d->epilogue, p.splitk > 1 ? " splitK" : "", d->gn_partial ? " +gn" : "", split ? " split3" : "");
    if (variant == 18) {
        GP_REQUIRE(smallm_ok, "gp_gemm: variant 18 needs a plain fp16 GEMM / conv (fp16 out, no split-K), N %% 32 == 0, M %% 16 == 0 (%% 64 with fused GroupNorm statistics)");
        GP_REQUIRE(d->M / (16 * sm_mt) <= 65535, "gp_gemm: variant 18: M=%d gives more than 65535 row tiles (grid.y)", d->M);
        if (d->gn_partial) {
            GP_REQUIRE(sm_mt == 1 || sm_mt == 2 || sm_mt == 4, "gp_gemm: variant 18: bad statistics chunk");
            if (d->KH > 0) { if (sm_mt == 4) launch_smallm<4, 2, true, true>(p, s); else if (sm_mt == 2) launch_smallm<2, 2, true, true>(p, s); else launch_smallm<1, 2, true, true>(p, s); }
            else { if (sm_mt == 4) launch_smallm<4, 2, false, true>(p, s); else if (sm_mt == 2) launch_smallm<2, 2, false, true>(p, s); else launch_smallm<1, 2, false, true>(p, s); }
        }
        else if (d->KH > 0) { if (sm_mt == 4) launch_smallm<4, 2, true, false>(p, s); else if (sm_mt == 2) launch_smallm<2, 2, true, false>(p, s); else launch_smallm<1, 2, true, false>(p, s); }
        else { if (sm_mt == 4) launch_smallm<4, 2, false, false>(p, s); else if (sm_mt == 2) launch_smallm<2, 2, false, false>(p, s); else launch_smallm<1, 2, false, false>(p, s); }
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 16) {
        GP_REQUIRE(wreg_ok, "gp_gemm: variant 16 needs a plain fp16 GEMM with K = 512, N %% 256 == 0, M %% 32 == 0, epilogue none/gelu/relu/lrelu");
        const int nsl = d->N / 256, tiles = d->M / 32;
        const int want = gp_num_cus() / nsl > 0 ? gp_num_cus() / nsl : 1;       // M groups so that one workgroup per CU covers the launch
        p.tiles_m = cdiv(tiles, want);                                            // tiles of 32 rows per group
        const int groups = cdiv(tiles, p.tiles_m);
        const int grid = cdiv(groups * nsl, 8) * 8;                               // padded items leave at once (T <= 0)
        switch (d->epilogue) {
            case GP_EPI_GELU:
#ifdef GP_WREG_STAMPS   // investigation build (GP_EXTRA_HIPCC_FLAGS=-DGP_WREG_STAMPS): variant 1616 = the same kernel with s_memtime
                        // stamps of workgroup 0 into the workspace (scripts/wreg_stamps.py); spills 7 registers -> never in the product
                if (p.dbg == 16 && p.ws) { hipLaunchKernelGGL((gemm_wreg_kernel<GP_EPI_GELU, 16>), dim3(grid), dim3(512), 0, s, p); break; }
                // timing ablations (wrong results): 1 no MFMA, 2 no stores, 4 no in-loop DMA, 8 no GELU in the MFMA shadow (and none at all), 10 = 8 + 2
                if (p.dbg == 1) { hipLaunchKernelGGL((gemm_wreg_kernel<GP_EPI_GELU, 1>), dim3(grid), dim3(512), 0, s, p); break; }
                if (p.dbg == 2) { hipLaunchKernelGGL((gemm_wreg_kernel<GP_EPI_GELU, 2>), dim3(grid), dim3(512), 0, s, p); break; }
                if (p.dbg == 4) { hipLaunchKernelGGL((gemm_wreg_kernel<GP_EPI_GELU, 4>), dim3(grid), dim3(512), 0, s, p); break; }
                if (p.dbg == 8) { hipLaunchKernelGGL((gemm_wreg_kernel<GP_EPI_NONE, 8>), dim3(grid), dim3(512), 0, s, p); break; }
                if (p.dbg == 10) { hipLaunchKernelGGL((gemm_wreg_kernel<GP_EPI_NONE, 10>), dim3(grid), dim3(512), 0, s, p); break; }
                if (p.dbg == 14) { hipLaunchKernelGGL((gemm_wreg_kernel<GP_EPI_NONE, 14>), dim3(grid), dim3(512), 0, s, p); break; }
#endif
                hipLaunchKernelGGL(gemm_wreg_kernel<GP_EPI_GELU>, dim3(grid), dim3(512), 0, s, p);
                break;
            case GP_EPI_RELU: hipLaunchKernelGGL(gemm_wreg_kernel<GP_EPI_RELU>, dim3(grid), dim3(512), 0, s, p); break;
            case GP_EPI_LRELU: hipLaunchKernelGGL(gemm_wreg_kernel<GP_EPI_LRELU>, dim3(grid), dim3(512), 0, s, p); break;
            default: hipLaunchKernelGGL(gemm_wreg_kernel<GP_EPI_NONE>, dim3(grid), dim3(512), 0, s, p); break;
        }
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 17) {
        GP_REQUIRE(wreg_ok && d->ldc % 8 == 0 && ((size_t)d->C & 15) == 0, "gp_gemm: variant 17 needs a plain fp16 GEMM with K = 512, N %% 256 == 0, M %% 32 == 0, ldc %% 8 == 0, "
                   "16-byte aligned C, epilogue none/gelu/relu/lrelu");
        const int nsl = d->N / 256, tiles = d->M / 32;
        const int want = gp_num_cus() / nsl > 0 ? gp_num_cus() / nsl : 1;
        p.tiles_m = cdiv(tiles, want);
        const int groups = cdiv(tiles, p.tiles_m);
        const int grid = cdiv(groups * nsl, 8) * 8;
        switch (d->epilogue) {
            case GP_EPI_GELU: hipLaunchKernelGGL(gemm_wreg2_kernel<GP_EPI_GELU>, dim3(grid), dim3(512), 0, s, p); break;
            case GP_EPI_RELU: hipLaunchKernelGGL(gemm_wreg2_kernel<GP_EPI_RELU>, dim3(grid), dim3(512), 0, s, p); break;
            case GP_EPI_LRELU: hipLaunchKernelGGL(gemm_wreg2_kernel<GP_EPI_LRELU>, dim3(grid), dim3(512), 0, s, p); break;
            default: hipLaunchKernelGGL(gemm_wreg2_kernel<GP_EPI_NONE>, dim3(grid), dim3(512), 0, s, p); break;
        }
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant >= 19 && variant <= 22) {   // weights in registers, two accumulator sets: 19 / 20 on 32x32x16 MFMAs, 22 / 21 on 16x16x32; 20 / 21: GELU on packed fp16
        GP_REQUIRE(wreg_ok && d->ldc % 8 == 0 && ((size_t)d->C & 15) == 0, "gp_gemm: variant 19-22 needs a plain fp16 GEMM with K = 512, N %% 256 == 0, M %% 32 == 0, ldc %% 8 == 0, "
                   "16-byte aligned C, epilogue none/gelu/relu/lrelu");
        const int nsl = d->N / 256, tiles = d->M / 32;
        const int want = gp_num_cus() / nsl > 0 ? gp_num_cus() / nsl : 1;
        p.tiles_m = cdiv(tiles, want);
        const int groups = cdiv(tiles, p.tiles_m);
        const int grid = cdiv(groups * nsl, 8) * 8;
        const bool s32 = variant <= 20, g16 = variant == 20 || variant == 21;
#define GP_W3(E, S, G) hipLaunchKernelGGL((gemm_wreg3_kernel<E, S, G>), dim3(grid), dim3(512), 0, s, p)
        switch (d->epilogue) {
            case GP_EPI_GELU:
                if (s32) { if (g16) GP_W3(GP_EPI_GELU, 1, 1); else GP_W3(GP_EPI_GELU, 1, 0); }
                else { if (g16) GP_W3(GP_EPI_GELU, 0, 1); else GP_W3(GP_EPI_GELU, 0, 0); }
                break;
            case GP_EPI_RELU: if (s32) GP_W3(GP_EPI_RELU, 1, 0); else GP_W3(GP_EPI_RELU, 0, 0); break;
            case GP_EPI_LRELU: if (s32) GP_W3(GP_EPI_LRELU, 1, 0); else GP_W3(GP_EPI_LRELU, 0, 0); break;
            default: if (s32) GP_W3(GP_EPI_NONE, 1, 0); else GP_W3(GP_EPI_NONE, 0, 0); break;
        }
#undef GP_W3
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 13) {
        GP_REQUIRE(win_ok, "gp_gemm: variant 13 needs a 3x3 s1 p1 fp16 conv with Cout 256, W in {64, 32, 16}");
        p.tiles_m = d->M / 256;
        p.tiles_n = 1;
        // wide tile (512 pixels x 128 channels) where it fits: bitwise the 256 x 256 tile's results, 2.3-3.5 % faster (scripts/conv_wide_ab.py,
        // profiles/r04_conv_wide_tile_ab.txt); dbg code 8 (variant 813) forces it, 9 (variant 913) / GP_CONV_WIDE=0 keep the square tile
        const bool wide = !split && d->Win >= 32 && d->M % 512 == 0 && d->H % (512 / d->Win) == 0 && p.dbg != 9 && (p.dbg == 8 || conv_wide_enabled());
        // 4-stage W ring, DMA lead 2 (a 5-stage / lead-3 instantiation spilled 43 registers and ran 1.6x slower)
        if (split) {
            if (d->Win == 64) hipLaunchKernelGGL((conv3_pp_kernel<64, 4, true>), dim3(p.tiles_m), dim3(512), 0, s, p);
            else if (d->Win == 32) hipLaunchKernelGGL((conv3_pp_kernel<32, 4, true>), dim3(p.tiles_m), dim3(512), 0, s, p);
            else hipLaunchKernelGGL((conv3_pp_kernel<16, 4, true>), dim3(p.tiles_m), dim3(512), 0, s, p);
        }
#ifdef GP_CONV_GNL   // investigation build only (GP_EXTRA_HIPCC_FLAGS=-DGP_CONV_GNL GP_BUILD_TAG=gnl): the arm spills 28 registers -- scratch is banned in the product
        else if (d->Win == 64 && p.dbg == 7) hipLaunchKernelGGL((conv3_pp_kernel<64, 4, false, true>), dim3(p.tiles_m), dim3(512), 0, s, p);
        else if (d->Win == 32 && p.dbg == 7) hipLaunchKernelGGL((conv3_pp_kernel<32, 4, false, true>), dim3(p.tiles_m), dim3(512), 0, s, p);
#endif
        else if (wide && d->Win == 64) { p.tiles_m = d->M / 512; hipLaunchKernelGGL((conv3_pp_kernel<64, 4, false, false, true>), dim3(p.tiles_m * 2), dim3(512), 0, s, p); }
        else if (wide && d->Win == 32) { p.tiles_m = d->M / 512; hipLaunchKernelGGL((conv3_pp_kernel<32, 4, false, false, true>), dim3(p.tiles_m * 2), dim3(512), 0, s, p); }
        else if (d->Win == 64) hipLaunchKernelGGL((conv3_pp_kernel<64, 4>), dim3(p.tiles_m), dim3(512), 0, s, p);
        else if (d->Win == 32) hipLaunchKernelGGL((conv3_pp_kernel<32, 4>), dim3(p.tiles_m), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv3_pp_kernel<16, 4>), dim3(p.tiles_m), dim3(512), 0, s, p);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 12) {   // as 10 with a 5-stage ring (all 160 KB of LDS), DMA lead 3
        if (d->dtype == GP_F16) launch_big<half_t, 2, 4, 8, 4, 5, false, 64, true>(p, s); else launch_big<float, 2, 2, 4, 4, 2>(p, s);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 10) {   // 256x256 ping-pong: 8 waves of 128x64, 64-byte K steps, 4-stage ring, one workgroup per CU
        if (split) launch_big<half_t, 2, 4, 8, 4, 4, false, 64, true, true>(p, s);
        else if (r32) launch_big<half_t, 2, 4, 8, 4, 4, false, 64, true, false, true>(p, s);
        else if (d->dtype == GP_F16) launch_big<half_t, 2, 4, 8, 4, 4, false, 64, true>(p, s); else launch_big<float, 2, 2, 4, 4, 2>(p, s);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 11) {   // 128x256 ping-pong: 8 waves of 64x64
        if (d->dtype == GP_F16) launch_big<half_t, 2, 4, 4, 4, 4, false, 64, true>(p, s); else launch_big<float, 2, 2, 4, 4, 2>(p, s);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 8) {   // 256x128, 4 waves of 128x64, 64-byte K steps, 3-stage ring: two workgroups per CU, so the
                          // epilogue of one overlaps the main loop of the other (short-K, store-heavy shapes)
        if (split) launch_big<half_t, 2, 2, 8, 4, 3, false, 64, false, true>(p, s);
        else if (d->dtype == GP_F16) launch_big<half_t, 2, 2, 8, 4, 3, false, 64>(p, s); else launch_big<float, 2, 2, 4, 4, 2>(p, s);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 9) {   // 128x128, 64-byte K steps, 4-stage ring, two workgroups per CU
        if (d->dtype == GP_F16) launch_big<half_t, 2, 2, 8, 4, 4, false, 64>(p, s); else launch_big<float, 2, 2, 4, 4, 2>(p, s);   // experiment: as 8 with 4 stages = 96 KB LDS = one workgroup per CU
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 7) {
        if (split) launch_big<half_t, 2, 2, 4, 4, 2, true, 128, false, true>(p, s);
        else if (r32) launch_big<half_t, 2, 2, 4, 4, 2, true, 128, false, false, true>(p, s);
        else if (d->dtype == GP_F16) launch_big<half_t, 2, 2, 4, 4, 2, true>(p, s); else launch_big<float, 2, 2, 4, 4, 2>(p, s);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 5) {
        if (d->dtype == GP_F16) launch_big<half_t, 2, 2, 4, 4, 4>(p, s); else launch_big<float, 2, 2, 4, 4, 4>(p, s);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 4) {
        if (p.splitk > 1) {
            // main kernel: raw fp32 partial sums, slab blockIdx.y of the workspace; then the reduce kernel with the real epilogue
            GemmKP q = p;
            q.C = d->workspace; q.out_f32 = 1; q.ldc = d->N; q.epi = GP_EPI_NONE; q.bias = nullptr; q.gamma = nullptr; q.res = nullptr;
            if (split) launch_big<half_t, 2, 2, 4, 4, 2, false, 128, false, true>(q, s);
            else if (d->dtype == GP_F16) launch_big<half_t, 2, 2, 4, 4, 2>(q, s); else launch_big<float, 2, 2, 4, 4, 2>(q, s);
            p.splitk = q.splitk;      // >= 2: the request was clamped to the number of K steps above
            const long work = (long)d->M * (d->N / 4);
            if (d->dtype == GP_F16 && !split) hipLaunchKernelGGL(splitk_reduce_kernel<half_t>, dim3(cdiv(work, 256)), dim3(256), 0, s, p);
            else hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(cdiv(work, 256)), dim3(256), 0, s, p);   // (split mode: fp32 residual / output)
            GP_LAUNCH_CHECK("gp_gemm");
        }
        if (split) launch_big<half_t, 2, 2, 4, 4, 2, false, 128, false, true>(p, s);
        else if (d->dtype == GP_F16) launch_big<half_t, 2, 2, 4, 4, 2>(p, s); else launch_big<float, 2, 2, 4, 4, 2>(p, s);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 3) {
        if (d->dtype == GP_F16) launch_big<half_t, 2, 4, 8, 4, 2>(p, s); else launch_big<float, 2, 2, 4, 4, 2>(p, s);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    if (variant == 2) {
        if (d->dtype == GP_F16) launch_big<half_t, 4, 2, 4, 4, 3>(p, s); else launch_big<float, 4, 2, 4, 4, 3>(p, s);
        GP_LAUNCH_CHECK("gp_gemm");
    }
    return gp_fail(GP_ERR_INVALID, "gp_gemm: variant %d not handled", variant);
}
