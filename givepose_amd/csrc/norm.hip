// Bandwidth-bound normalisation kernels on channels-last tensors (gfx950):
//   depth-wise KxK conv + LayerNorm (+GELU), row LayerNorm, GroupNorm statistics / apply.
// Common mapping: a pixel's C channels are spread over CT = C / VEC consecutive threads, VEC = one
// 16-byte vector (8 halfs / 4 floats), so every global access is a full 16 B per lane and a pixel row
// is one contiguous run per CT lanes; 256-thread blocks hold PG = 256 / CT pixel groups.
#include "common.hpp"

#ifdef GP_DW_STAMPS   // investigation build: s_memtime stamps of ONE workgroup of dwconv7_ln_mfma_kernel (scripts/dw_stamps.py)
__device__ unsigned long long gp_dw_stamp_buf[8 * 16];
extern "C" int gp_dw_stamps_read(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(gp_dw_stamp_buf), sizeof(unsigned long long) * 8 * 16) == hipSuccess ? 0 : 1;
}
#define GP_DW_MARK(k) do { if (blockIdx.x == gridDim.x / 2 + 1 && blockIdx.y == 0) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) gp_dw_stamp_buf[(threadIdx.x >> 6) * 16 + (k)] = t__; } } while (0)
__device__ unsigned long long gp_dwt_stamp_buf[8 * 64];       // dwconv7_ln_tall_kernel (scripts/dw_tall_stamps.py)
extern "C" int gp_dwt_stamps_read(unsigned long long* host) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(gp_dwt_stamp_buf), sizeof(unsigned long long) * 8 * 64) == hipSuccess ? 0 : 1;
}
#define GP_DWT_MARK(k) do { if (blockIdx.x == gridDim.x / 2 + 1) { const unsigned long long t__ = __builtin_amdgcn_s_memtime(); if ((threadIdx.x & 63) == 0) gp_dwt_stamp_buf[(threadIdx.x >> 6) * 64 + (k)] = t__; } } while (0)
#else
#define GP_DW_MARK(k) do { } while (0)
#define GP_DWT_MARK(k) do { } while (0)
#endif
namespace {

// Sum NV values over the CT threads (CT power of two, 16..256, aligned) that share this thread's
// pixel group.  red: LDS scratch of 4*NV floats.
template <int NV>
__device__ __forceinline__ void pixel_group_sum(float (&v)[NV], int CT, float* red) {
    if (CT <= 64) {
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = group_sum(v[i], CT);
    } else {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i] = group_sum(v[i], 64);
        __syncthreads();
        if (lane == 0)
#pragma unroll
            for (int i = 0; i < NV; ++i) red[wave * NV + i] = v[i];
        __syncthreads();
        const int wpg = CT >> 6, w0 = (threadIdx.x / CT) * wpg;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float s = 0.f;
            for (int w = 0; w < wpg; ++w) s += red[(w0 + w) * NV + i];
            v[i] = s;
        }
    }
}

template <typename T> __device__ __forceinline__ void load_f32(const float* p, float* o);
template <> __device__ __forceinline__ void load_f32<half_t>(const float* p, float* o) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    for (int i = 0; i < 4; ++i) { o[i] = a[i]; o[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load_f32<float>(const float* p, float* o) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p);
    for (int i = 0; i < 4; ++i) o[i] = a[i];
}

// acc[e] += in[e] * w[e] over one 16-byte vector.  f16: v_fma_mix_f32 takes the two fp16 operands straight from
// the packed registers (fp32 accumulate), 1 VALU op per MAC instead of cvt + cvt + fma: the depth-wise kernels are
// VALU bound once their traffic is on-chip.
template <typename T> __device__ __forceinline__ void mac16(float* acc, const Vec16<T>& a, const Vec16<T>& w);
template <> __device__ __forceinline__ void mac16<float>(float* acc, const Vec16<float>& a, const Vec16<float>& w) {
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = fmaf(a.e[e], w.e[e], acc[e]);
}
template <> __device__ __forceinline__ void mac16<half_t>(float* acc, const Vec16<half_t>& a, const Vec16<half_t>& w) {
    const unsigned* au = reinterpret_cast<const unsigned*>(&a.u);
    const unsigned* wu = reinterpret_cast<const unsigned*>(&w.u);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * j]) : "v"(au[j]), "v"(wu[j]));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(acc[2 * j + 1]) : "v"(au[j]), "v"(wu[j]));
    }
}

// ---------------------------------------------------------------------------- dwconv + LN (+act)
#ifndef DW_STRIP_PD
#define DW_STRIP_PD 2
#endif
// PPT pixels (one strip of a row) per thread: 8 for throughput; 2 where 8 would leave most of the chip idle (the detections of
// one frame: 256 pixels of a stage-2 map are 8 workgroups at PPT = 8 and 32 at PPT = 2 -- more halo reads, all of them cache hits)
template <typename T, int KS, int PPT>
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const T* __restrict__ x, const T* __restrict__ wt,
                                                        const float* __restrict__ bias,
                                                        const float* __restrict__ lnw,
                                                        const float* __restrict__ lnb, T* __restrict__ y, int H,
                                                        int W, int C, float eps, int act, long n_pixels, long pl,
                                                        const int* __restrict__ grp = nullptr) {
    constexpr int VEC = Vec16<T>::N, R = KS / 2;
    __shared__ float red[4 * PPT];
    const int CT = C / VEC, PG = 256 / CT;
    const int cs = threadIdx.x % CT, pg = threadIdx.x / CT;
    const long strip = (long)blockIdx.x * PG + pg;
    const long pix0 = strip * PPT;
    const bool valid = pix0 < n_pixels;
    // grp (gp_dwconv_ln_groups): the output is the quarter-size flat prefix list of SEVERAL batches behind one another (H W / 4 rows per
    // crop); output row j of a crop whose batch starts at crop g reads the flat full-resolution pixel j + 3 g (H W / 4) -- row j - g q of
    // that batch's own flat list.  g is clamped to [0, crop]: whatever the table holds, the source stays inside x.
    long pix0s = pix0;
    if (grp != nullptr && valid) {
        const long q = (long)H * W / 4, c = pix0 / q;
        const long g = min((long)max(grp[c], 0), c);
        pix0s += 3 * g * q;
    }
    const int w0 = (int)(pix0s % W);
    const long t = pix0s / W;
    const int h = (int)(t % H);
    const long b = t / H;
    float acc[PPT][VEC];
    {
        float bv[VEC];
        load_f32<T>(bias + cs * VEC, bv);
#pragma unroll
        for (int p = 0; p < PPT; ++p)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[p][e] = bv[e];
    }
    if constexpr (PPT == 2) {
        // latency form: the input pixels and taps of filter rows kh + 1 .. kh + PD are requested before row kh is multiplied (rolling
        // window: the seven load -> multiply rounds of the loop below overlap instead of queueing up).  In a hipGraph chain at 1 crop:
        // 9.5 -> 8.3 us per launch (PD = 1); inside the network PD = 2 is another 1-2 % of a 4-crop forward, PD = 3 nothing more;
        // requesting the WHOLE 7 x 8 window first was slower (13.4 us)
        if (valid) {
            const T* xb = x + (b * H * W) * C + cs * VEC;
            constexpr int PD = DW_STRIP_PD;      // filter rows in flight ahead of the one being multiplied
            Vec16<T> in[PD + 1][KS + 1], wv[PD + 1][KS];
            auto fetch = [&](int kh, Vec16<T> (&dst)[KS + 1], Vec16<T> (&wd)[KS]) {
                const int hi = h + kh - R;
                const bool rowok = (unsigned)hi < (unsigned)H;
#pragma unroll
                for (int c = 0; c < KS + 1; ++c) {
                    const int wi = w0 + c - R;
                    dst[c] = (rowok && (unsigned)wi < (unsigned)W) ? load16<T>(xb + ((long)hi * W + wi) * C) : zero16<T>();
                }
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) wd[kw] = load16<T>(wt + (long)(kh * KS + kw) * C + cs * VEC);
            };
#pragma unroll
            for (int k0 = 0; k0 < PD && k0 < KS; ++k0) fetch(k0, in[k0 % (PD + 1)], wv[k0 % (PD + 1)]);
#pragma unroll
            for (int kh = 0; kh < KS; ++kh) {
                if (kh + PD < KS) fetch(kh + PD, in[(kh + PD) % (PD + 1)], wv[(kh + PD) % (PD + 1)]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kw = 0; kw < KS; ++kw) {
                    mac16<T>(acc[0], in[kh % (PD + 1)][kw], wv[kh % (PD + 1)][kw]);
                    mac16<T>(acc[1], in[kh % (PD + 1)][kw + 1], wv[kh % (PD + 1)][kw]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (valid) {
        const T* xb = x + (b * H * W) * C + cs * VEC;
        for (int kh = 0; kh < KS; ++kh) {
            const int hi = h + kh - R;
            if ((unsigned)hi >= (unsigned)H) continue;
            Vec16<T> in[KS + PPT - 1];
#pragma unroll
            for (int c = 0; c < KS + PPT - 1; ++c) {
                const int wi = w0 + c - R;
                in[c] = ((unsigned)wi < (unsigned)W) ? load16<T>(xb + ((long)hi * W + wi) * C) : zero16<T>();
            }
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const Vec16<T> wv = load16<T>(wt + (long)(kh * KS + kw) * C + cs * VEC);
#pragma unroll
                for (int p = 0; p < PPT; ++p) mac16<T>(acc[p], in[p + kw], wv);
            }
        }
    }
    // LayerNorm over C per pixel (two-pass: mean, then centred variance)
    float s[PPT];
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        s[p] = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) s[p] += acc[p][e];
    }
    pixel_group_sum<PPT>(s, CT, red);
    float v[PPT];
    const float invC = 1.0f / C;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const float mean = s[p] * invC;
        v[p] = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            acc[p][e] -= mean;
            v[p] += acc[p][e] * acc[p][e];
        }
    }
    pixel_group_sum<PPT>(v, CT, red);
    if (!valid) return;
    float gw[VEC], gb[VEC];
    load_f32<T>(lnw + cs * VEC, gw);
    load_f32<T>(lnb + cs * VEC, gb);
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        if (pix0 + p >= n_pixels) break;
        const float rstd = rsqrtf(v[p] * invC + eps);
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float val = acc[p][e] * rstd * gw[e] + gb[e];
            o.set(e, act == GP_ACT_GELU ? gelu_for<T>(val) : apply_act(val, act));
        }
        store16p<T>(y, (pix0 + p) * C + cs * VEC, o, pl);
    }
}

// ---------------------------------------------------------------------------- dw3x3 + LN (+GELU), LDS-tiled (round 6)
// The DCNv3 prefix kernel (ops_dcnv3/modules/dcnv3.py:318-356: x1 = GELU(LN(dw3x3(x))), C = 256) ran on the strip kernel above at 2.7-3.4 x its HBM floor
// (64 x 64 prefix of 64 crops: 28.9 us for 67 MB, profiles/r06_dw3_forms.txt): every thread fetched its own 3 x 4 input window and its nine taps from L1 / L2,
// row by row.  Here a workgroup owns 16 x TH output pixels x all 256 channels: the 18 x (TH + 2) halo window (36 KB at TH = 2, 54 KB at 4; out-of-image pixels are
// zeros) goes through LDS once, a thread (channel octet o = t % 32, pixel slot t / 32: 16 / PPT slots per tile row) convolves PPT = 2 TH neighbouring pixels of one
// row from PPT + 2 LDS vectors per filter row.  Arithmetic, its order, the two-pass LayerNorm (group_sum over the pixel's 32 lanes) and the GELU are the strip
// kernel's: same bits.  Only whole TH-row blocks of the flat (image, row) list (H % TH == 0, so a block never straddles two images; the host checks n_pixels).
// GELU only (the DCNv3 prefix kernel's activation, compiled in: the strip kernel's per-element `act` switch costs a scalar branch per value -- 2 896 -> 2 210
// vector instructions per thread; without an activation the compiler contracts the LayerNorm's last multiply-add differently in the two kernels (1 ulp), so
// that case stays on the strip kernel)
template <int TH>      // tile height: 4 (8 pixels per thread, 54 KB of LDS: two workgroups per CU) or 2 (4 pixels per thread, 36 KB: four)
__global__ __launch_bounds__(256) void dwconv3_ln_tile_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wt, const float* __restrict__ bias,
                                                              const float* __restrict__ lnw, const float* __restrict__ lnb, half_t* __restrict__ y,
                                                              int H, int W, float eps) {
    constexpr int C = 256, HR = TH + 2, HC = 18, PPT = 2 * TH, SPR = 16 / PPT;      // pixels per thread, pixel slots per tile row
    extern __shared__ __attribute__((aligned(16))) char dw3_lds[];      // [HR][HC][C] halfs = 36 864 (TH = 2) / 55 296 (TH = 4) bytes
    const int t = threadIdx.x, o = t & 31, slot = t >> 5, r = slot / SPR, ch = slot % SPR;
    const int tpr = W >> 4;
    const int rb = blockIdx.x / tpr, c0 = (blockIdx.x - rb * tpr) * 16;
    const long row0 = (long)rb * TH;                // first row of the block in the flat (image, row) list
    const int h0 = (int)(row0 % H);
    const half_t* xim = x + (row0 - h0) * W * C;    // pixel (0, 0) of the block's image
    // ---- halo window -> LDS: piece i = (halo pixel i / 32, octet i % 32); 32 neighbouring lanes fetch one pixel's 512 contiguous bytes
    constexpr int NPC = HR * HC * 32, NIT = (NPC + 255) / 256;
    uint4 pc[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = t + 256 * k, px = i >> 5, hr = px / HC, hc = px - hr * HC;
        const int hi = h0 - 1 + hr, wi = c0 - 1 + hc;
        pc[k] = (i < NPC && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W)
                    ? *reinterpret_cast<const uint4*>(xim + ((long)hi * W + wi) * C + (i & 31) * 8) : uint4{0u, 0u, 0u, 0u};
    }
    Vec16<half_t> wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = load16<half_t>(wt + k * C + o * 8);
    float acc[PPT][8];
    {
        float bv[8];
        load_f32<half_t>(bias + o * 8, bv);
#pragma unroll
        for (int p = 0; p < PPT; ++p)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[p][e] = bv[e];
    }
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int i = t + 256 * k;
        if (i < NPC) *reinterpret_cast<uint4*>(dw3_lds + (size_t)i * 16) = pc[k];
    }
    __syncthreads();
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        Vec16<half_t> in[PPT + 2];
#pragma unroll
        for (int j = 0; j < PPT + 2; ++j) in[j].u = *reinterpret_cast<const uint4*>(dw3_lds + (size_t)(((r + kh) * HC + ch * PPT + j) * 32 + o) * 16);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int p = 0; p < PPT; ++p) mac16<half_t>(acc[p], in[p + kw], wv[kh * 3 + kw]);
    }
    // ---- LayerNorm over the pixel's 256 channels = its 32 lanes (two-pass, as the strip kernel), activation, store
    float sm[PPT], vr[PPT];
    const float invC = 1.0f / C;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) a += acc[p][e];
        sm[p] = group_sum(a, 32);
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const float mean = sm[p] * invC;
        float a = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            acc[p][e] -= mean;
            a += acc[p][e] * acc[p][e];
        }
        vr[p] = group_sum(a, 32);
    }
    float gw[8], gb[8];
    load_f32<half_t>(lnw + o * 8, gw);
    load_f32<half_t>(lnb + o * 8, gb);
    half_t* yb = y + ((row0 + r) * W + c0 + ch * PPT) * C + o * 8;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        const float rstd = rsqrtf(vr[p] * invC + eps);
        Vec16<half_t> ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float val = acc[p][e] * rstd * gw[e] + gb[e];
            ov.set(e, gelu_for<half_t>(val));
        }
        store16<half_t>(yb + (long)p * C, ov);
    }
}

// ---------------------------------------------------------------------------- dw7x7 + LN, LDS-tiled
// The strip kernel above re-reads every input row 7x and every filter tap once per 8 pixels from L1/L2 and is
// bound by L2->L1 traffic (12x read amplification, profiles/r01a).  Here a workgroup owns a TW x TH output tile
// of one image and walks the channels in slabs of 16 lanes x 16 B: the (TW+6) x (TH+6) halo tile of a slab and
// its 49 taps are brought in by LDS-DMA (zero page for the padding) into one of two LDS buffers while the
// previous slab is being convolved out of the other; accumulators of all slabs stay in registers until the
// LayerNorm statistics over C are known.  Halo amplification: 3.1x (8x8 tile), all of it L2 hits.

__device__ __forceinline__ void glds16_n(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

template <typename T, int NSLAB>
__global__ __launch_bounds__(512) void dwconv7_ln_tiled_kernel(const T* __restrict__ x, const T* __restrict__ wt,
                                                               const float* __restrict__ bias,
                                                               const float* __restrict__ lnw,
                                                               const float* __restrict__ lnb, T* __restrict__ y, int H,
                                                               int W, int C, float eps, int dbg, long pl) {
    constexpr int VEC = Vec16<T>::N, SC = 16 * VEC, KS = 7, R = 3;
    constexpr int PPT = 2, SPR = 4;                  // 32 pixel-threads x 2 px: strips per tile row
    constexpr int TW = SPR * PPT, TH = 32 / SPR;     // 8 x 8 output tile, 8 waves (2 per SIMD)
    constexpr int IW = TW + 6, IH = TH + 6, NPX = IW * IH;
    constexpr int IN_INSTR = (NPX * 16 + 63) / 64, W_INSTR = (49 * 16 + 63) / 64;
    constexpr int BUF = (IN_INSTR + W_INSTR) * 1024;
    constexpr int NBUF = NSLAB > 1 ? 2 : 1;
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    typedef __attribute__((address_space(3))) char lds_char_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)dsm;
    // bias / LN weight / LN bias (3 x C fp32) are staged in LDS once: a global load inside the slab loop would make
    // hipcc wait vmcnt(0), i.e. for the in-flight LDS-DMA of the next slab, serialising DMA and convolution.
    const float* par_s = reinterpret_cast<const float*>(dsm + NBUF * BUF);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tpr = W / TW, tpi = tpr * (H / TH);
    const int b = blockIdx.x / tpi, tin = blockIdx.x - b * tpi;
    const int h0 = (tin / tpr) * TH, w0 = (tin % tpr) * TW;
    const T* xb = x + (long)b * H * W * C;
    const T* zero = reinterpret_cast<const T*>(gp_zero_page_tu);

    auto issue = [&](int s, int buf) {
        const unsigned base = lds0 + buf * BUF;
        for (int ins = wave; ins < IN_INSTR; ins += 8) {
            const int i = ins * 64 + lane, px = i >> 4, sl = i & 15;
            const int iy = px / IW, ix = px - iy * IW;
            const int gy = h0 - R + iy, gx = w0 - R + ix;
            const T* src = zero;
            if (px < NPX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)
                src = xb + ((long)gy * W + gx) * C + s * SC + sl * VEC;
            glds16_n(src, base + ins * 1024);
        }
        for (int ins = wave; ins < W_INSTR; ins += 8) {
            const int i = ins * 64 + lane, tap = i >> 4, sl = i & 15;
            const T* src = tap < 49 ? wt + (long)tap * C + s * SC + sl * VEC : zero;
            glds16_n(src, base + (IN_INSTR + ins) * 1024);
        }
    };

    const int slot = tid & 15, pt = tid >> 4;
    const int row = pt / SPR, col0 = (pt % SPR) * PPT;
    float acc[NSLAB][PPT][VEC];
    {
        const int pin = (3 * C * 4 + 1023) / 1024;   // 1 KB DMA pieces of the parameter block
        for (int ins = wave; ins < pin; ins += 8) {
            const int f = (ins * 64 + lane) * 4;      // float index into [bias | lnw | lnb]
            const float* src = f < C ? bias + f : (f < 2 * C ? lnw + (f - C) : (f < 3 * C ? lnb + (f - 2 * C) : reinterpret_cast<const float*>(gp_zero_page_tu)));
            glds16_n(src, lds0 + NBUF * BUF + ins * 1024);
        }
    }
    if (dbg != 2) issue(0, 0);
#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s + 1 < NSLAB && dbg != 2) issue(s + 1, (s + 1) & 1);
        const char* in_s = dsm + (s & (NBUF - 1)) * BUF;
        const char* w_s = in_s + IN_INSTR * 1024;
        {
            float bv[VEC];
            load_f32<T>(par_s + s * SC + slot * VEC, bv);
#pragma unroll
            for (int p = 0; p < PPT; ++p)
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[s][p][e] = bv[e];
        }
#pragma unroll 1
        for (int kh = 0; kh < (dbg == 1 ? 0 : KS); ++kh) {
            Vec16<T> in[KS + PPT - 1];
#pragma unroll
            for (int c = 0; c < KS + PPT - 1; ++c)
                in[c].u = *reinterpret_cast<const uint4*>(in_s + (((row + kh) * IW + col0 + c) * 16 + slot) * 16);
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                Vec16<T> wv;
                wv.u = *reinterpret_cast<const uint4*>(w_s + ((kh * KS + kw) * 16 + slot) * 16);
#pragma unroll
                for (int p = 0; p < PPT; ++p) mac16<T>(acc[s][p], in[p + kw], wv);
            }
        }
        if (NBUF == 1) __syncthreads();
    }
    // LayerNorm over C: channels of a pixel live in NSLAB registers-slabs x 16 lanes
    float sum[PPT], var[PPT];
    const float invC = 1.0f / C;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
            for (int e = 0; e < VEC; ++e) a += acc[s][p][e];
        sum[p] = group_sum(a, 16) * invC;
    }
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                acc[s][p][e] -= sum[p];
                a += acc[s][p][e] * acc[s][p][e];
            }
        var[p] = rsqrtf(group_sum(a, 16) * invC + eps);
    }
#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
        float gw[VEC], gb[VEC];
        load_f32<T>(par_s + C + s * SC + slot * VEC, gw);
        load_f32<T>(par_s + 2 * C + s * SC + slot * VEC, gb);
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.set(e, acc[s][p][e] * var[p] * gw[e] + gb[e]);
            store16p<T>(y, (((long)b * H + h0 + row) * W + w0 + col0 + p) * C + s * SC + slot * VEC, o, pl);
        }
    }
}

template <typename T, int NSLAB>
void launch_dw7_tiled(const void* x, const void* wt, const float* bias, const float* lnw, const float* lnb, void* y, int B,
                      int H, int W, int C, float eps, hipStream_t s, int dbg = 0, long pl = 0) {
    constexpr int NPX = 14 * 14, IN_INSTR = (NPX * 16 + 63) / 64, W_INSTR = (49 * 16 + 63) / 64;
    const int LDS = (NSLAB > 1 ? 2 : 1) * (IN_INSTR + W_INSTR) * 1024 + ((3 * C * 4 + 1023) / 1024) * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)dwconv7_ln_tiled_kernel<T, NSLAB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((dwconv7_ln_tiled_kernel<T, NSLAB>), dim3(B * (H / 8) * (W / 8)), dim3(512), LDS, s, (const T*)x,
                       (const T*)wt, bias, lnw, lnb, (T*)y, H, W, C, eps, dbg, pl);
}

// ---------------------------------------------------------------------------- dw7x7 + LN on the matrix cores (fp16)
// The tiled kernel above is VALU bound (49 taps x 8 channels of v_fma_mix per pixel-thread).  A depth-wise filter is a
// block-diagonal matrix product, and the block structure is a free choice.  Round 1-3: one output row, 16 channels per MFMA
//     D[c][px] += sum_{(tap j, c')} A[c][(j, c')] B[(j, c')][px],  A = diag over c of w[tap j][c],  two taps (kh, kh + 1) per MFMA
// -- 1/16 of the MACs useful, 3.5 MFMAs per output row of 16 channels x 16 pixels.  Round 4 (this form): TWO output rows x
// EIGHT channels per MFMA and a K block of FOUR input rows x 8 channels,
//     D[(r, c)][px] += sum_{(i, c')} A[(r, c)][(i, c')] B[(i, c')][px],   A[(r, c)][(i, c')] = (c == c') w[kh = i + 4 blk - r][kw][c]
//     (zero where kh is outside 0..6),  B[(i, c')][px] = in[row 2 p + 4 blk + i][px + kw][c'],
// i.e. a banded (Toeplitz over the rows) x diagonal (over the channels) A: the output rows 2p, 2p + 1 need the input rows
// 2p .. 2p + 7 = exactly two K blocks, 14 of whose 16 (r, i) pairs carry a tap.  Two MFMAs per row PAIR of 8 channels x 16
// pixels = 2 per output row of 16 channels, against 3.5: 0.57 of the matrix-pipe time, which is what bounds these kernels once
// their traffic is on chip (C = 128 at 64 x 64: 57 us of MFMA issue per launch of 128 crops in the old form).
// A lane of a B fragment reads ONE 16-byte slot (8 channels of one pixel of one input row: lane = (pixel n, row i)); bits 1-3 of
// a pixel's slot index are XOR-ed with its halo column and bit 0 with its halo ROW parity, so that the lanes of every
// ds_read_b128 lane group ({0-3,12-15,20-27}, ...: two input rows of opposite parity, eight column residues each) hit 16
// different slots for all 7 column shifts.  An A fragment is one non-zero half per lane, built from the tap table in LDS.
// A wave owns 16 channels (two octets) per 128-channel slab; output tile 16 x 4 pixels (two row pairs), slabs by LDS-DMA
// (double-buffered for C = 512 on small grids, single-buffered at two workgroups per CU otherwise).
// What bounds it (scripts/dw_ablate.py, profiles/r04_dw_*): not the matrix pipe any more -- with the conv loop AND the halo DMA
// removed the kernel keeps 50-55 % of its time, and a PERSISTENT form (one workgroup per CU walking a run of tiles, taps and
// parameters resident, the next item's halo landing while this one is convolved / normalised / stored; built, bitwise equal,
// profiles/r04_dw_persistent_ab.txt, removed again) ran no faster than two one-tile workgroups per CU: 55 KB of halo in flight per
// CU against ~2 us of memory latency is what sets the pace (13-18 GB/s per CU), not launch or index overheads.
// RAW = true: one workgroup per (tile, 128-channel slab) (blockIdx.y = slab), no LayerNorm: y gets the conv + bias
// output rounded to fp16 and `stats` (pixel, {sum, sum of squares}, slab) the per-pixel partial moments of those
// ROUNDED values over the slab's channels; the LayerNorm is applied by the consuming GEMM's epilogue
// (GP_EPI_LNFOLD_GELU).  Removes the serial slab chain of a workgroup and the all-channel reduction.
template <int NSLAB, int NBUF, bool RAW = false>
__global__ __launch_bounds__(512, NBUF == 1 ? 2 : 1) void dwconv7_ln_mfma_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wt,
                                                              const float* __restrict__ bias,
                                                              const float* __restrict__ lnw,
                                                              const float* __restrict__ lnb, half_t* __restrict__ y,
                                                              int H, int W, int C, float eps, int dbg,
                                                              float* __restrict__ stats = nullptr) {
    static_assert(!RAW || (NSLAB == 1 && NBUF == 1), "raw mode: one slab per workgroup");
    const int slab0 = RAW ? (int)blockIdx.y : 0;
    constexpr int TW = 16, TH = 4, R = 3, IW = TW + 6, IH = TH + 6, NPX = IW * IH;
    constexpr int IN_INSTR = (NPX * 16 + 63) / 64, W_INSTR = (49 * 16 + 63) / 64;
    constexpr int BUF = (IN_INSTR + W_INSTR) * 1024;   // halo tile directly followed by the taps (see row 10 below)
    constexpr bool PIPE = NBUF == 2;   // two workgroups per CU hide the LDS latency by themselves
    constexpr int ROWB = IW * 256;
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    typedef __attribute__((address_space(3))) char lds_char_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)dsm;
    const int par_bytes = ((3 * C * 4 + 1023) / 1024) * 1024;
    const float* par_s = reinterpret_cast<const float*>(dsm + NBUF * BUF);
    float* red_s = reinterpret_cast<float*>(dsm + NBUF * BUF + par_bytes);   // [8 waves][64 px] (+ the same again for the squares)
    float* stat_s = red_s + 8 * 64;                                          // RAW: [64 px] + the squares' partials behind it

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    GP_DW_MARK(0);
    const int tpr = W / TW, tpi = tpr * (H / TH);
    const int bid = xcd_chunk(blockIdx.x, gridDim.x);   // tiles of an XCD = a contiguous run of rows (common.hpp)
    const int b = bid / tpi, tin = bid - b * tpi;
    const int h0 = (tin / tpr) * TH, w0 = (tin % tpr) * TW;
    const half_t* xb = x + (long)b * H * W * C;
    const half_t* zero = reinterpret_cast<const half_t*>(gp_zero_page_tu);

    // DMA sources of slab 0, computed once: the halo pixel of a lane does not depend on the slab (only + 256 B per slab)
    constexpr int XPW = (IN_INSTR + 7) / 8, WPW = (W_INSTR + 7) / 8;
    const half_t* xsrc0[XPW];
    const half_t* wsrc0[WPW];
    unsigned xok = 0, wok = 0;
#pragma unroll
    for (int j = 0; j < XPW; ++j) {
        const int ins = wave + 8 * j;
        const int i = ins * 64 + lane, px = i >> 4, ps = i & 15;
        const int iy = px / IW, ix = px - iy * IW;
        const int gy = h0 - R + iy, gx = w0 - R + ix;
        const int ls = ps ^ ((ix & 7) << 1) ^ (iy & 1);
        const bool ok = ins < IN_INSTR && px < NPX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        xsrc0[j] = ok ? xb + ((long)gy * W + gx) * C + slab0 * 128 + ls * 8 : zero;
        if (ok) xok |= 1u << j;
    }
#pragma unroll
    for (int j = 0; j < WPW; ++j) {
        const int ins = wave + 8 * j;
        const int i = ins * 64 + lane, tap = i >> 4, sl = i & 15;
        const bool ok = ins < W_INSTR && tap < 49;
        wsrc0[j] = ok ? wt + (long)tap * C + slab0 * 128 + sl * 8 : zero;
        if (ok) wok |= 1u << j;
    }
    auto issue = [&](int s, int buf) {
        const unsigned base = lds0 + buf * BUF;
#pragma unroll
        for (int j = 0; j < XPW; ++j)
            if (wave + 8 * j < IN_INSTR) glds16_n((xok >> j) & 1 ? xsrc0[j] + s * 128 : zero, base + (wave + 8 * j) * 1024);
#pragma unroll
        for (int j = 0; j < WPW; ++j)
            if (wave + 8 * j < W_INSTR) glds16_n((wok >> j) & 1 ? wsrc0[j] + s * 128 : zero, base + (IN_INSTR + wave + 8 * j) * 1024);
    };
    {
        const int pin = par_bytes / 1024;
        for (int ins = wave; ins < pin; ins += 8) {
            const int f = (ins * 64 + lane) * 4;
            const float* src = f < C ? bias + f : (f < 2 * C ? lnw + (f - C) : (f < 3 * C ? lnb + (f - 2 * C) : reinterpret_cast<const float*>(gp_zero_page_tu)));
            glds16_n(src, lds0 + NBUF * BUF + ins * 1024);
        }
    }
    if (dbg != 6 && dbg != 8) issue(0, 0);
    GP_DW_MARK(1);

    // lane roles.  B / D column: output pixel n of the row; q = lane >> 4: as a B lane the input row of the K block, as a D lane
    // the rows 4q .. 4q + 3 of D = output row 2p + (q >> 1), channels 8 o + 4 (q & 1) .. + 4 of the wave's octet o.  As an A
    // lane: row (ar, ac) = (output row of the pair, channel of the octet) = (n >> 3, n & 7), K chunk q = input row of the block.
    const int n = lane & 15, q = lane >> 4;
    const int ar = n >> 3, ac = n & 7;
    unsigned sw[7];   // byte offset of this lane's B slot (octet 0; octet 1 = ^ 16) for shift kw, halo row q
#pragma unroll
    for (int kw = 0; kw < 7; ++kw)
        sw[kw] = q * ROWB + (n + kw) * 256 + ((((wave * 2) ^ (((n + kw) & 7) << 1)) ^ (q & 1)) << 4);
    // tap row of this A lane: kh = q - ar (K block 0, valid from 0 up) / q + 4 - ar (block 1, valid up to 6); an invalid lane reads a
    // clamped tap and is masked to zero.  The lane's one non-zero half sits in dword ac >> 1, half ac & 1.
    const int kh0 = q - ar, kh1 = q + 4 - ar;
    const unsigned woff0 = (unsigned)((kh0 < 0 ? 0 : kh0) * 7 * 256 + (wave * 16 + ac) * 2);
    const unsigned woff1 = (unsigned)((kh1 > 6 ? 6 : kh1) * 7 * 256 + (wave * 16 + ac) * 2);
    unsigned mk0[4], mk1[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        mk0[v] = (kh0 >= 0 && v == (ac >> 1)) ? 0xffffffffu : 0u;
        mk1[v] = (kh1 <= 6 && v == (ac >> 1)) ? 0xffffffffu : 0u;
    }
    const int sh = 16 * (ac & 1);
    constexpr int NP = TH / 2;   // output row pairs of the tile

    f32x4 acc[NSLAB][NP][2];    // [slab][row pair][octet]
#pragma unroll
    for (int s = 0; s < NSLAB; ++s)
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int o = 0; o < 2; ++o) acc[s][pp][o] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int s = 0; s < NSLAB; ++s) {
        if (s == 0) GP_DW_MARK(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (s == 0) GP_DW_MARK(3);
        __syncthreads();
        if (s == 0) GP_DW_MARK(4);
        if (NBUF == 2 && s + 1 < NSLAB && dbg != 6 && dbg != 8) issue(s + 1, (s + 1) & 1);
        const char* in_s = dsm + (s & (NBUF - 1)) * BUF;
        const char* w_s = in_s + IN_INSTR * 1024;
        union Frag { uint4 u; half8 h; };
        // software pipeline over the 7 column shifts (PIPE): the 4 tap words and 4 NP B fragments of shift kw+1 are fetched
        // while the 4 NP MFMAs of shift kw run.  Fragment (pp, blk, o): halo rows 2 pp + 4 blk .. + 4 (lane: + q), octet o.
        Frag af[PIPE ? 2 : 1][2][2], bf[PIPE ? 2 : 1][NP][2][2];    // af[..][blk][o], bf[..][pp][blk][o]
        unsigned wraw[2][2];
        auto fetch = [&](int kw, int slot) {   // LDS reads only: the VALU part (build) must not sit in front of the MFMAs
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                wraw[0][o] = *reinterpret_cast<const unsigned short*>(w_s + woff0 + kw * 256 + o * 16);
                wraw[1][o] = *reinterpret_cast<const unsigned short*>(w_s + woff1 + kw * 256 + o * 16);
            }
#pragma unroll
            for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int o = 0; o < 2; ++o)
                        bf[slot][pp][blk][o].u = *reinterpret_cast<const uint4*>(in_s + (sw[kw] ^ (o << 4)) + (2 * pp + 4 * blk) * ROWB);
        };
        auto build = [&](int slot) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const unsigned u0 = wraw[0][o] << sh, u1 = wraw[1][o] << sh;
                af[slot][0][o].u = uint4{u0 & mk0[0], u0 & mk0[1], u0 & mk0[2], u0 & mk0[3]};
                af[slot][1][o].u = uint4{u1 & mk1[0], u1 & mk1[1], u1 & mk1[2], u1 & mk1[3]};
            }
        };
        if (PIPE) { fetch(0, 0); build(0); }
#pragma unroll
        for (int kw = 0; kw < (dbg == 5 || dbg == 8 || dbg == 9 ? 0 : 7); ++kw) {   // timing ablations: 5 no conv, 6 no halo DMA, 8 neither, 9 no conv + no output
            if (!PIPE) { fetch(kw, 0); build(0); }
            else if (kw + 1 < 7) fetch(kw + 1, (kw + 1) & 1);
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int pp = 0; pp < NP; ++pp)
#pragma unroll
                    for (int o = 0; o < 2; ++o)
                        acc[s][pp][o] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[PIPE ? kw & 1 : 0][blk][o].h, bf[PIPE ? kw & 1 : 0][pp][blk][o].h, acc[s][pp][o], 0, 0, 0);
            if (PIPE) {
                __builtin_amdgcn_sched_barrier(0);
                if (kw + 1 < 7) build((kw + 1) & 1);
            }
        }
        if (NBUF == 1 && s + 1 < NSLAB) {
            __syncthreads();
            if (dbg != 6 && dbg != 8) issue(s + 1, 0);
        }
    }

    GP_DW_MARK(5);
    // D layout: lane (n, q) holds, of slab s, row pair pp and octet o, the channels s 128 + wave 16 + 8 o + 4 (q & 1) + {0..3} of
    // pixel (row 2 pp + (q >> 1), n).  Bias, then LayerNorm over C.
    const int rsel = q >> 1;
    const int cq = wave * 16 + (q & 1) * 4;      // + 8 o
#pragma unroll
    for (int s = 0; s < NSLAB; ++s)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float4 bv = *reinterpret_cast<const float4*>(par_s + (slab0 + s) * 128 + cq + o * 8);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                acc[s][pp][o][0] += bv.x; acc[s][pp][o][1] += bv.y; acc[s][pp][o][2] += bv.z; acc[s][pp][o][3] += bv.w;
            }
        }
    if constexpr (RAW) {
        float* red2_s = stat_s + 64;                 // [8 waves][64 px] sums of squares
        half4 ov[NP][2];
#pragma unroll
        for (int pp = 0; pp < NP; ++pp) {
            float a = 0.f, a2 = 0.f;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    ov[pp][o][e] = (_Float16)acc[0][pp][o][e];
                    const float f = (float)ov[pp][o][e];
                    a += f;
                    a2 += f * f;
                }
            a += __shfl_xor(a, 16); a2 += __shfl_xor(a2, 16);      // the other channel quad of the octets (q ^ 1)
            if ((q & 1) == 0) { const int px = (2 * pp + rsel) * 16 + n; red_s[wave * 64 + px] = a; red2_s[wave * 64 + px] = a2; }
        }
        __syncthreads();   // also: every wave is done reading the halo tile, which the output tile overlays
        const int nsl = C >> 7;
        if (tid < 128) {
            const int which = tid >> 6, px = tid & 63;
            const float* r = which ? red2_s : red_s;
            float a = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) a += r[w8 * 64 + px];
            const long pixel = ((long)b * H + h0 + (px >> 4)) * W + w0 + (px & 15);
            stats[(pixel * 2 + which) * nsl + slab0] = a;
        }
        char* out_r = dsm;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int o = 0; o < 2; ++o)
                *reinterpret_cast<half4*>(out_r + ((2 * pp + rsel) * 16 + n) * 256 + (((wave * 2 + o) ^ n) << 4) + (q & 1) * 8) = ov[pp][o];
        __syncthreads();
        for (int i = tid; i < 64 * 16; i += 512) {
            const int px = i >> 4, c = i & 15;
            const uint4 v = *reinterpret_cast<const uint4*>(out_r + px * 256 + ((c ^ (px & 15)) << 4));
            *reinterpret_cast<uint4*>(y + (((long)b * H + h0 + (px >> 4)) * W + w0 + (px & 15)) * C + slab0 * 128 + c * 8) = v;
        }
        return;
    }
    // LayerNorm statistics in ONE round: per-wave partial (sum, sum of squares) of every pixel -> LDS, one barrier, then every
    // lane adds the eight partials of ITS pixels itself (broadcast reads).  var = E[x^2] - mean^2 in fp32: the conv outputs have
    // |mean| of the order of their spread, so the subtraction costs a few of fp32's 24 bits -- nothing at fp16 output precision.
    // (Until round 4: two passes (mean, then centred squares), each with a 64-thread reduction stage = four barriers per tile;
    // the tile is 64 pixels and the kernel is made of such fixed costs: scripts/dw_ablate.py.)
    const float invC = 1.0f / C;
    float mean[NP], rstd[NP];
    float* red2_s = red_s + 8 * 64;              // [8 waves][64 px] sums of squares (behind the sums: launch_dw7_mfma sizes both)
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        float a = 0.f, a2 = 0.f;
#pragma unroll
        for (int s = 0; s < NSLAB; ++s)
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a += acc[s][pp][o][e];
                    a2 += acc[s][pp][o][e] * acc[s][pp][o][e];
                }
        a += __shfl_xor(a, 16);
        a2 += __shfl_xor(a2, 16);
        if ((q & 1) == 0) { red_s[wave * 64 + (2 * pp + rsel) * 16 + n] = a; red2_s[wave * 64 + (2 * pp + rsel) * 16 + n] = a2; }
    }
    GP_DW_MARK(6);
    __syncthreads();
    GP_DW_MARK(7);
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        float a = 0.f, a2 = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) { a += red_s[w8 * 64 + (2 * pp + rsel) * 16 + n]; a2 += red2_s[w8 * 64 + (2 * pp + rsel) * 16 + n]; }
        mean[pp] = a * invC;
        rstd[pp] = rsqrtf(fmaxf(a2 * invC - mean[pp] * mean[pp], 0.f) + eps);
    }
    // normalised rows go through LDS (the DMA buffers are free now) so that the global stores are 16 B per lane and
    // 1 KB contiguous per pixel; 16-byte chunks of a pixel are XOR-swizzled with the pixel index.
    char* out_s = dsm;
#pragma unroll
    for (int s = 0; s < NSLAB; ++s)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float4 gw = *reinterpret_cast<const float4*>(par_s + C + s * 128 + cq + o * 8);
            const float4 gb = *reinterpret_cast<const float4*>(par_s + 2 * C + s * 128 + cq + o * 8);
            const int chunk = s * 16 + wave * 2 + o;
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) {
                half4 ov;
                ov[0] = (_Float16)((acc[s][pp][o][0] - mean[pp]) * rstd[pp] * gw.x + gb.x);
                ov[1] = (_Float16)((acc[s][pp][o][1] - mean[pp]) * rstd[pp] * gw.y + gb.y);
                ov[2] = (_Float16)((acc[s][pp][o][2] - mean[pp]) * rstd[pp] * gw.z + gb.z);
                ov[3] = (_Float16)((acc[s][pp][o][3] - mean[pp]) * rstd[pp] * gw.w + gb.w);
                const int px = (2 * pp + rsel) * 16 + n;
                *reinterpret_cast<half4*>(out_s + px * (C * 2) + ((chunk ^ n) << 4) + (q & 1) * 8) = ov;
            }
        }
    GP_DW_MARK(8);
    __syncthreads();
    GP_DW_MARK(9);
    const int cpp = C / 8;   // 16-byte chunks per pixel
    for (int i = tid; i < 64 * cpp; i += 512) {
        const int px = i / cpp, c = i - px * cpp;
        const uint4 v = *reinterpret_cast<const uint4*>(out_s + px * (C * 2) + ((c ^ (px & 15)) << 4));
        const int t = px >> 4, mm = px & 15;
        if (dbg != 9 || v.x == 0x12345678u)
            *reinterpret_cast<uint4*>(y + (((long)b * H + h0 + t) * W + w0 + mm) * C + c * 8) = v;
    }
    GP_DW_MARK(10);
#ifdef GP_DW_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GP_DW_MARK(11);
#endif
}

template <int NSLAB, int NBUF>
void launch_dw7_mfma(const void* x, const void* wt, const float* bias, const float* lnw, const float* lnb, void* y, int B,
                     int H, int W, int C, float eps, hipStream_t s, int dbg) {
    constexpr int NPX = 22 * 10, IN_INSTR = (NPX * 16 + 63) / 64, W_INSTR = (49 * 16 + 63) / 64;
    const int LDS = NBUF * (IN_INSTR + W_INSTR) * 1024 + ((3 * C * 4 + 1023) / 1024) * 1024 + 16 * 64 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)dwconv7_ln_mfma_kernel<NSLAB, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((dwconv7_ln_mfma_kernel<NSLAB, NBUF>), dim3(B * (H / 4) * (W / 16)), dim3(512), LDS, s, (const half_t*)x,
                       (const half_t*)wt, bias, lnw, lnb, (half_t*)y, H, W, C, eps, dbg);
}

void launch_dw7_raw(const void* x, const void* wt, const float* bias, void* y, float* stats, int B, int H, int W, int C,
                    hipStream_t s) {
    constexpr int NPX = 22 * 10, IN_INSTR = (NPX * 16 + 63) / 64, W_INSTR = (49 * 16 + 63) / 64;
    const int LDS = (IN_INSTR + W_INSTR) * 1024 + ((3 * C * 4 + 1023) / 1024) * 1024 + 17 * 64 * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)dwconv7_ln_mfma_kernel<1, 1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((dwconv7_ln_mfma_kernel<1, 1, true>), dim3(B * (H / 4) * (W / 16), C / 128), dim3(512), LDS, s, (const half_t*)x,
                       (const half_t*)wt, bias, bias, bias, (half_t*)y, H, W, C, 0.f, 0, stats);
}

// ---------------------------------------------------------------------------- dw7x7 + LN, 16-pixel-wide maps: 16 x 8 tiles (round 6)
// ConvNeXt stage 2 (16 x 16 x 512: 27 of the 36 blocks) is one round of workgroups whatever the tiling, so a workgroup's own chain
// IS the launch: the 16 x 4 kernel above spends 33 of its 52 k cycles convolving four 128-channel slabs one after the other, each behind
// its own DMA, at ~45 % of the matrix pipe's rate (profiles/r04_dw_stamps.txt), and fetches every input row 2.1 x (10 halo rows per 4).
// This form:
//   * HALF an image (16 columns x 8 rows) x ALL channels per workgroup: 11 real input rows per 8 (1.375 x), no column halo at all -- the
//     columns left / right of the map and the rows above / below it are ZERO PIXELS THAT LIVE IN LDS (two per row, whole rows outside the
//     map), written once per workgroup; an out-of-map lane of a B fragment reads the zero pixel of its row;
//   * 64-channel slabs through a THREE-stage LDS-DMA ring (input rows + the slab's 49 taps: 38.5 KB per stage): slab s + 2 is requested
//     when slab s starts, so a slab has two conv periods to land and only the first one is waited for;
//   * a wave owns ONE channel octet of every slab and FOUR output row pairs: each tap fragment (A operand, built by VALU) feeds four
//     MFMAs instead of two, and the six distinct 4-row B fragments of a column shift feed eight MFMAs (row pairs p and p + 2 share
//     rows 2 p + 4 .. 2 p + 7): 0.75 LDS reads per MFMA, so the loop is bound by the matrix pipe, not by the LDS port (256 B / clk);
//   * the accumulators of all slabs (128 registers at C = 512) stay in registers; LayerNorm statistics in one round as above; the
//     normalised half image (128 KB at C = 512) is staged in LDS once -- the ring is dead by then -- and leaves as 1 KB rows.
// LDS image of a stage: pixel slot P = r * 18 + cc (r: 14 halo rows, cc: 16 columns + zero pixels 16 / 17), 128 B per pixel = 8 slots
// of 16 B (one channel octet each); octet o of column c sits in slot o ^ ((c >> 1) & 7) and an out-of-map column c reads zero pixel
// 16 + (c & 1) at the slot its own c asks for: the 16 lanes of every ds_read_b128 lane group (8 columns of two adjacent rows) then hit
// 16 different 16-byte bank groups for all seven column shifts (enumerated: scripts/probes/dw_tall_swizzle.py).  Taps: 128 B per tap,
// 16-byte chunk c of filter row kh stored at chunk c ^ kh (the A lanes of a ds_read_u16 come from up to seven filter rows).
__device__ __forceinline__ void glds16_sv(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

// WIDE (maps wider than 16 pixels: ConvNeXt stages 0 / 1): the same kernel on 16 x 8 tiles WITH a column halo -- a stage holds the 14 x 22 pixels around
// the tile (pixel slot P = r * 22 + cc, the same octet swizzle: 16 consecutive columns of two adjacent rows still hit 16 different bank groups), loaded
// in pieces of 8 consecutive slots; a lane whose slot lies outside the map is EXEC-masked out of the LDS-DMA instruction and its slot keeps the zero the
// workgroup wrote once (a piece with no lane inside the map goes to the padding KB instead, so that every slab is the same number of instructions for the
// counted vmcnt waits).  NS = 2 (C = 128): both slabs resident (two stages); NS = 4: the three-stage ring.
#define GP_WAIT_VMCNT_J(J) do { if constexpr ((J) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else { static_assert((J) == 9, "J"); asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); } } while (0)
// TH = 4 (16-wide maps only): quarter-image tiles for the launches whose half-image tiles would leave half the chip idle (33 .. 64 crops at stage 2: the
// strictly serial bs-64 forward): two row pairs per wave -- the B-fragment sharing is gone (4 fragments for 4 MFMAs per column shift) -- everything else as above.
// PAIR (8 x 8 maps: ConvNeXt stage 3, C = 1024): TWO images side by side in one 16-column tile -- lane n of a B fragment is column n & 7 of image n >> 3, the pixel
// slots of a row are [image A: 8][image B: 8][zero][zero] and a column outside its OWN image reads a zero pixel (the same swizzle stays conflict-free:
// scripts/probes/dw_tall_swizzle.py); TH = 4 (128 accumulator registers at 16 slabs) or, the routed form, TH = 2 (one row pair per wave: B fragments 0 and 2 only,
// one staging half in the epilogue; 2 B workgroups).  The images of a pair are consecutive in memory.
template <int NS, int J, bool WIDE = false, bool NOMFMA = false, int TH = 8, bool PAIR = false>    // C = 64 NS channels; J LDS-DMA instructions per LOADING wave (waves 0-3) and slab (4 J >= input pieces + 7); NOMFMA: timing ablation (wrong results)
__global__ __launch_bounds__(512, WIDE ? 2 : 1) void dwconv7_ln_tall_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wt,
                                                                 const float* __restrict__ bias, const float* __restrict__ lnw,
                                                                 const float* __restrict__ lnb, half_t* __restrict__ y, int H, int Wrt, float eps) {
    constexpr int C = 64 * NS, NP = TH / 2, IH = TH + 6, PITCH = WIDE ? 22 : 18, ROWB = PITCH * 128, NPX = 16 * TH, NBF = NP + 2;
    static_assert(TH == 8 || (TH == 4 && !WIDE) || (TH == 2 && PAIR), "tile height");
    static_assert(!PAIR || (!WIDE && (TH == 4 || TH == 2)), "pair tiles: 8-wide maps, half- or quarter-image tiles of two images");
    constexpr int NHF = NP >= 2 ? 2 : 1, PPH = NP / NHF;       // halves of the staging epilogue, row pairs per half
    constexpr int IN_BYTES = (IH * ROWB + 1023) / 1024 * 1024, TAP_OFF = IN_BYTES, PAD_OFF = TAP_OFF + 7 * 1024, STAGE = PAD_OFF + 1024;   // PAD_OFF: 1 KB, target of the padding DMA instructions
    // WIDE: ONE stage and two workgroups per CU (<= 128 registers, 57-59 KB of LDS each): a map of 64 x 64 is many rounds of tiles, so it is another
    // workgroup's conv that covers this one's prologue, slab latency and epilogue, not a ring inside the workgroup (measured: one workgroup per CU with a
    // ring ran 131 us at C = 128 against the 16 x 4 kernel's 119: profiles/r06_dw_tall_ab.txt)
    constexpr int NST = WIDE ? 1 : (NS >= 3 ? 3 : NS);
    constexpr int NIN_W = (IH * PITCH + 7) / 8;                // WIDE: input pieces of 8 pixel slots per slab (39)
    const int W = WIDE ? Wrt : 16;
    constexpr int OUT_BYTES = NPX * C * 2;                     // the normalised tile, staged over the dead ring
    constexpr int PAR_OFF = (OUT_BYTES > NST * STAGE ? OUT_BYTES : NST * STAGE), PAR_INS = (3 * C * 4 + 1023) / 1024;
    constexpr int RED_OFF = PAR_OFF + PAR_INS * 1024;          // [2][8 waves][128 px] fp32
    static_assert(RED_OFF + 2 * 8 * NPX * 4 <= 160 * 1024, "LDS");
    static_assert(PAR_INS <= 16, "at most two parameter DMA instructions per wave");
    extern __shared__ __attribute__((aligned(1024))) char dsm[];
    typedef __attribute__((address_space(3))) char lds_char_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)dsm;
    const float* par_s = reinterpret_cast<const float*>(dsm + PAR_OFF);
    float* red_s = reinterpret_cast<float*>(dsm + RED_OFF);
    float* red2_s = red_s + 8 * NPX;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tpr = W / 16, tpi = (H / TH) * tpr;
    const int bid = xcd_chunk(blockIdx.x, gridDim.x);
    const int b = bid / tpi, tin = bid - b * tpi, h0 = (tin / tpr) * TH, w0 = (tin - (tin / tpr) * tpr) * 16;
    const half_t* xb = x + (long)b * H * W * C;
    // halo rows r = 0 .. 13 are image rows h0 - 3 + r; the real ones are rlo .. rhi
    const int rlo = h0 >= 3 ? 0 : 3 - h0, rhi = min(IH - 1, H - 1 - (h0 - 3)), nreal = rhi - rlo + 1;
    const int nin = WIDE ? NIN_W : 2 * nreal;                  // input pieces (LDS-DMA instructions) per slab
    GP_DWT_MARK(0);

    // ---- DMA plan.  ALL LDS-DMA is issued by waves 0-3 (one per SIMD): an LDS-DMA instruction holds its wave for 100-250 cycles (the CU's
    //      address path takes 16 cycles per KB and every loading wave queues behind the others), cycles in which that wave issues no MFMA;
    //      the hardware favours the older wave of a SIMD, so waves 0-3 finish their MFMAs ~700 cycles before waves 4-7 anyway and used to
    //      spend them at the slab barrier (profiles/r06_dw_tall_stamps.txt).  Instruction slots i = wave + 4 j of a slab: 2 x nreal input
    //      half rows, then 7 KB of taps, then padding.  First thing in the kernel: everything else runs under the first slabs' latency.
    constexpr int NLW = WIDE ? 8 : 4;      // loading waves (WIDE: all of them -- another workgroup of the CU multiplies meanwhile, and 6 instead of 12 offset registers keep the wave under 128)
    const bool loader = wave < NLW;
    auto slot_plan = [&](int i, unsigned long long& sb, unsigned& vo, unsigned& ds, bool& ok) {     // i: scalar slot index of a slab's DMA list; ok: this lane takes part
        ok = true;
        if (WIDE && i < nin) {
            const int P = i * 8 + (lane >> 3), r = (P * 745) >> 14, cc = P - r * PITCH, o = (lane & 7) ^ ((cc >> 1) & 7);     // P / 22 for P < 312
            const int gy = h0 - 3 + r, gx = w0 - 3 + cc;
            ok = r < IH && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
            sb = (unsigned long long)xb;
            vo = ok ? (unsigned)(((gy * W + gx) * C + o * 8) * 2) : 0u;
            ds = (unsigned)(i * 1024);
            if (__builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(ok) == 0 ? 1 : 0)) {     // nobody inside the map: a padding piece (the instruction count per slab stays fixed)
                ok = true;
                vo = (unsigned)((lane & 15) * 16);
                ds = PAD_OFF;
            }
        } else if (!WIDE && i < nin) {
            const int r = rlo + (i >> 1), cc = (i & 1) * 8 + (lane >> 3), o = (lane & 7) ^ ((cc >> 1) & 7);
            sb = (unsigned long long)xb;
            vo = PAIR ? (unsigned)((((i & 1) * H * 8 + (h0 - 3 + r) * 8 + (lane >> 3)) * C + o * 8) * 2)      // piece (r, half) = row r of image `half` of the pair
                      : (unsigned)((((h0 - 3 + r) * W + cc) * C + o * 8) * 2);
            ds = (unsigned)((r * PITCH + (i & 1) * 8) * 128);
        } else if (i < nin + 7) {
            const int t7 = i - nin, tap = min(t7 * 8 + (lane >> 3), 48), kh = (tap * 37) >> 8, ch = (lane & 7) ^ kh;   // tap / 7 for tap < 56
            sb = (unsigned long long)wt;
            vo = (unsigned)((tap * C + ch * 8) * 2);
            ds = (unsigned)(TAP_OFF + t7 * 1024);
        } else {
            sb = (unsigned long long)xb;
            vo = (unsigned)((lane & 15) * 16);
            ds = PAD_OFF;
        }
    };
    // slab 0 and the parameters by ALL eight waves, first thing in the kernel (the launch is one round of workgroups: nothing hides this
    // latency); slabs 1 and 2 follow from the loading waves behind the first barrier
#pragma unroll
    for (int ins = wave; ins < PAR_INS; ins += 8) {
        const int f = (ins * 64 + lane) * 4;
        const float* src = f < C ? bias + f : (f < 2 * C ? lnw + (f - C) : lnb + (f - 2 * C));
        glds16_n(f < 3 * C ? (const void*)src : (const void*)gp_zero_page_tu, __builtin_amdgcn_readfirstlane(lds0 + PAR_OFF + ins * 1024));
    }
#pragma unroll
    for (int j = 0; j < (WIDE ? J : (J + 1) / 2); ++j) {
        const int i = wave + 8 * j;
        if (i < nin + 7) {
            unsigned long long sb; unsigned vo, ds; bool ok;
            slot_plan(i, sb, vo, ds, ok);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)sb), hi = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
            const unsigned dd = __builtin_amdgcn_readfirstlane(lds0 + ds);
            if (ok) glds16_sv((const void*)(((unsigned long long)hi << 32) | lo), vo, dd);
        }
    }
    unsigned voff[J];
    unsigned vbits = 0;                    // bit j: this lane takes part in piece j (WIDE: lanes whose pixel slot lies outside the map do not)
    unsigned slo[J], shi[J], sdst[J];      // wave-uniform source base of slab 0 (+ 128 bytes per slab: input and taps alike; padding re-reads input row 0) and LDS target
#pragma unroll
    for (int j = 0; j < J; ++j) {
        unsigned long long sb; unsigned ds; bool ok;
        slot_plan((wave & (NLW - 1)) + NLW * j, sb, voff[j], ds, ok);
        vbits |= ok ? 1u << j : 0u;
        slo[j] = __builtin_amdgcn_readfirstlane((unsigned)sb);
        shi[j] = __builtin_amdgcn_readfirstlane((unsigned)(sb >> 32));
        sdst[j] = __builtin_amdgcn_readfirstlane(lds0 + ds);
    }
    auto issue1 = [&](int s, int j) {      // DMA instruction j of slab s (s, j compile-time at every call site)
        const unsigned long long sbu = (((unsigned long long)shi[j] << 32) | slo[j]) + (unsigned)(s * 128);
        if constexpr (WIDE) { if ((vbits >> j) & 1) glds16_sv((const void*)sbu, voff[j], sdst[j] + (s % NST) * STAGE); }
        else glds16_sv((const void*)sbu, voff[j], sdst[j] + (s % NST) * STAGE);
    };
    auto issue = [&](int s) {
#pragma unroll
        for (int j = 0; j < J; ++j) issue1(s, j);
    };
    GP_DWT_MARK(1);

    // ---- zero pixels (two per halo row) and the halo rows outside the map, all stages: written once, never touched by the DMA
    if constexpr (WIDE) {
        // every 16-byte unit of a stage's pixel slots that lies outside the map (edge tiles only; interior tiles write nothing)
        for (int u = tid; u < NIN_W * 64; u += 512) {
            const int P = u >> 3, r = (P * 745) >> 14, cc = P - r * PITCH;
            const int gy = h0 - 3 + r, gx = w0 - 3 + cc;
            if (!(r < IH && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W)) {
#pragma unroll
                for (int st = 0; st < NST; ++st) *reinterpret_cast<uint4*>(dsm + st * STAGE + u * 16) = uint4{0u, 0u, 0u, 0u};
            }
        }
    } else {
        const int r = tid >> 4, u = tid & 15;
        if (r < IH) {
#pragma unroll
            for (int st = 0; st < NST; ++st) *reinterpret_cast<uint4*>(dsm + st * STAGE + (r * PITCH + 16) * 128 + u * 16) = uint4{0u, 0u, 0u, 0u};
        }
        const int nbad = IH - nreal;                           // <= 6 rows of 128 16-byte units (the 16 real pixel slots)
        for (int rr = tid >> 7; rr < nbad; rr += 4) {
            const int rz = rr < rlo ? rr : rhi + 1 + (rr - rlo);
#pragma unroll
            for (int st = 0; st < NST; ++st) *reinterpret_cast<uint4*>(dsm + st * STAGE + rz * ROWB + (tid & 127) * 16) = uint4{0u, 0u, 0u, 0u};
        }
    }

    // ---- lane roles (as dwconv7_ln_mfma_kernel): B / D column n = pixel column, q = input row of the K block; A lane: row (ar, ac), K chunk q
    const int n = lane & 15, q = lane >> 4;
    const int ar = n >> 3, ac = n & 7;
    unsigned sw[7];       // byte offset of this lane's B slot for column shift kw (row block 0): row q, column n + kw - 3
#pragma unroll
    for (int kw = 0; kw < 7; ++kw) {
        if constexpr (PAIR) {
            const int c = (n & 7) + kw - 3, cs = (n >> 3) * 8 + c;      // column of the lane's own image; slot in the row
            const int cc = (unsigned)c < 8u ? cs : 16 + (cs & 1);
            sw[kw] = (unsigned)(q * ROWB + cc * 128 + (((wave ^ (cs >> 1)) & 7) << 4));
        } else {
        const int c = WIDE ? n + kw : n + kw - 3;      // WIDE: halo column
        const int cc = (WIDE || (unsigned)c < 16u) ? c : 16 + (c & 1);
        sw[kw] = (unsigned)(q * ROWB + cc * 128 + (((wave ^ (c >> 1)) & 7) << 4));
        }
    }
    const int kh0 = q - ar, kh1 = q + 4 - ar;
    const int kh0c = kh0 < 0 ? 0 : kh0, kh1c = kh1 > 6 ? 6 : kh1;
    const unsigned woff0 = (unsigned)(TAP_OFF + kh0c * 7 * 128 + ((wave ^ kh0c) << 4) + ac * 2);
    const unsigned woff1 = (unsigned)(TAP_OFF + kh1c * 7 * 128 + ((wave ^ kh1c) << 4) + ac * 2);
    unsigned mk0[4], mk1[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        mk0[v] = (kh0 >= 0 && v == (ac >> 1)) ? 0xffffffffu : 0u;
        mk1[v] = (kh1 <= 6 && v == (ac >> 1)) ? 0xffffffffu : 0u;
    }
    const int sh = 16 * (ac & 1);
    // D layout: lane (n, q), slab s, row pair pp: channels 64 s + 8 wave + 4 (q & 1) + {0..3} of pixel (row 2 pp + (q >> 1), column n)
    const int rsel = q >> 1, cq = wave * 8 + (q & 1) * 4;
    GP_DWT_MARK(2);

    f32x4 acc[NS][NP];
    union Frag { uint4 u; half8 h; };
    constexpr bool AF1 = WIDE && NS >= 4;          // (C = 256 at two workgroups per CU: ONE set of A fragments, built in front of its MFMAs: 8 registers)
    Frag af[AF1 ? 1 : 2][2], bf[2][NBF];           // af[slot][blk]; bf[slot][row block rb / 2]: halo rows rb .. rb + 3, rb = 0, 2, .., 2 (NP + 1); slot = (7 s + kw) & 1
    constexpr bool PACKW = WIDE && NS >= 4;        // two taps per register (7 instead of 14): C = 256 at two workgroups per CU has 128 registers
    unsigned wraw[7][PACKW ? 1 : 2];               // the 14 taps of this lane's A rows (filter rows kh0 / kh1, all column shifts): read once per slab
    auto fetch = [&](const char* in_s, int kw, int slot) {
#pragma unroll
        for (int rb = 0; rb < NBF; ++rb)
            if (NP > 1 || rb != 1) bf[slot][rb].u = *reinterpret_cast<const uint4*>(in_s + sw[kw] + 2 * rb * ROWB);      // (one row pair: fragments 0 and 2 only)
    };
    auto taps = [&](const char* in_s) {
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
            const unsigned t0 = *reinterpret_cast<const unsigned short*>(in_s + woff0 + kw * 128);
            const unsigned t1 = *reinterpret_cast<const unsigned short*>(in_s + woff1 + kw * 128);
            if constexpr (PACKW) wraw[kw][0] = t0 | (t1 << 16);
            else { wraw[kw][0] = t0; wraw[kw][PACKW ? 0 : 1] = t1; }
        }
    };
    auto build = [&](int kw, int slot) {
        const unsigned u0 = (PACKW ? wraw[kw][0] & 0xffffu : wraw[kw][0]) << sh, u1 = (PACKW ? wraw[kw][0] >> 16 : wraw[kw][PACKW ? 0 : 1]) << sh;
        af[AF1 ? 0 : slot][0].u = uint4{u0 & mk0[0], u0 & mk0[1], u0 & mk0[2], u0 & mk0[3]};
        af[AF1 ? 0 : slot][1].u = uint4{u1 & mk1[0], u1 & mk1[1], u1 & mk1[2], u1 & mk1[3]};
    };
    // ONE barrier per slab, in front of the MFMAs of its LAST column shift: by then every fragment of slab s is in registers (stage s is
    // free: slab s + 3 goes into it) and the loading waves have seen slab s + 1 land (slab s + 2 stays in flight: every slab has two
    // conv periods to land); behind it the operands of slab s + 1's first column shift are fetched under the eight MFMAs that are left,
    // so that the matrix pipe does not drain at a slab boundary (a barrier FOLLOWED by the first LDS round trip cost ~700 cycles per slab)
    auto publish = [&](auto sc) {                  // slab s is done being read by this wave; make slab s + 1 visible
        constexpr int s = decltype(sc)::value;
        if constexpr (NST == 1) {                  // one stage: everybody is done reading slab s, THEN slab s + 1 is fetched into the same stage, waited for and published
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (loader) {
                issue(s + 1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            return;
        }
        if (loader) {
            if constexpr (s + 2 < NS && NST == 3) GP_WAIT_VMCNT_J(J);
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the fragment reads of slab s (and, s = -1, the zero fill)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if constexpr (s + NST < NS) {
            if (loader) issue(s + NST);
        }
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // own pieces of slab 0 (and of the parameters)
    GP_DWT_MARK(4);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    GP_DWT_MARK(5);
    taps(dsm);
    fetch(dsm, 0, 0);
    if (loader) {
        if (NST > 1) issue(1);
        if (NST > 2) issue(2);
    }
    if constexpr (!AF1) build(0, 0);
    static_for<0, NS>([&](auto sc) {
        constexpr int s = decltype(sc)::value;
        const char* in_s = dsm + (s % NST) * STAGE;
        {   // the bias is the accumulators' initial value
            const float4 bv = *reinterpret_cast<const float4*>(par_s + s * 64 + cq);
#pragma unroll
            for (int pp = 0; pp < NP; ++pp) acc[s][pp] = f32x4{bv.x, bv.y, bv.z, bv.w};
        }
        GP_DWT_MARK(6 + 4 * s);
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {           // operands one column shift ahead: the reads and the A-fragment VALU work of shift kw + 1 ride BETWEEN the MFMAs of shift kw
            const int slot = (7 * s + kw) & 1;     //   (an MFMA holds the issue port for 8 of its 16 cycles: in program order behind it, 1-2 other instructions are free)
            if (kw == 6 && s + 1 < NS && NST > 1) {
                GP_DWT_MARK(8 + 4 * s);
                publish(sc);
                GP_DWT_MARK(9 + 4 * s);
                const char* nx = dsm + ((s + 1) % NST) * STAGE;
                taps(nx);                          // wraw is dead: build(6) ran with the MFMAs of shift 5
                fetch(nx, 0, slot ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kw + 1 < 7) { fetch(in_s, kw + 1, slot ^ 1); if constexpr (!AF1) build(kw + 1, slot ^ 1); }
            if constexpr (AF1) build(kw, 0);
            if constexpr (!NOMFMA) {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp)
                        acc[s][pp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[AF1 ? 0 : slot][blk].h, bf[slot][pp + 2 * blk].h, acc[s][pp], 0, 0, 0);
            } else {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int pp = 0; pp < NP; ++pp) acc[s][pp][0] += __builtin_bit_cast(float, af[AF1 ? 0 : slot][blk].u.x ^ bf[slot][pp + 2 * blk].u.x);   // keeps the operand work alive
            }
            if (kw + 1 < 7 && !NOMFMA && !AF1) {
                static_for<0, 2 * NP>([&](auto ic) {    // MFMA, a fragment read (the first NP + 2), VALU instructions (10 per column shift)
                    constexpr int i = decltype(ic)::value;
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if constexpr (i < NBF) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if constexpr (NP == 4) { if constexpr (i < 2) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 1, 0); }
                    else { if constexpr (i < 2) __builtin_amdgcn_sched_group_barrier(0x002, 3, 0); else __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
                });
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (s + 1 < NS && NST == 1) {
            publish(sc);
            taps(dsm);
            fetch(dsm, 0, (7 * (s + 1)) & 1);
        }
        if constexpr (s + 1 < NS && !AF1) build(0, (7 * (s + 1)) & 1);     // first A fragments of the next slab (its taps were requested behind the barrier)
        GP_DWT_MARK(7 + 4 * s);
    });

    // ---- LayerNorm statistics (one round), normalise, stage, store
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        float a = 0.f, a2 = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a += acc[s][pp][e]; a2 = fmaf(acc[s][pp][e], acc[s][pp][e], a2); }
        a += __shfl_xor(a, 16);
        a2 += __shfl_xor(a2, 16);
        if ((q & 1) == 0) { const int px = (2 * pp + rsel) * 16 + n; red_s[wave * NPX + px] = a; red2_s[wave * NPX + px] = a2; }
    }
    GP_DWT_MARK(40);
    __syncthreads();      // also: every wave is done with the last slab's stage, which the output tile overlays
    GP_DWT_MARK(41);
    const float invC = 1.0f / C;
    float rstd[NP], nmr[NP];     // (v - mean) rstd = fma(v, rstd, -mean rstd)
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        const int px = (2 * pp + rsel) * 16 + n;
        float a = 0.f, a2 = 0.f;
#pragma unroll
        for (int w8 = 0; w8 < 8; ++w8) { a += red_s[w8 * NPX + px]; a2 += red2_s[w8 * NPX + px]; }
        const float mean = a * invC;
        rstd[pp] = rsqrtf(fmaxf(a2 * invC - mean * mean, 0.f) + eps);
        nmr[pp] = -mean * rstd[pp];
    }
    char* out_s = dsm;
    constexpr int cpp = C / 8;   // 16-byte chunks per pixel
    half_t* yb = PAIR ? y + ((long)b * 2 * H + h0) * 8 * C : y + (((long)b * H + h0) * W + w0) * C;      // PAIR: b = pair index
    // two halves (TH = 8: row pairs 0-1 / 2-3 = pixels 0-63 / 64-127): the stores of the first leave while the second is normalised
#pragma unroll
    for (int hf = 0; hf < NHF; ++hf) {
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float4 gw = *reinterpret_cast<const float4*>(par_s + C + s * 64 + cq);
            const float4 gb = *reinterpret_cast<const float4*>(par_s + 2 * C + s * 64 + cq);
            const int chunk = s * 8 + wave;
#pragma unroll
            for (int p2 = 0; p2 < PPH; ++p2) {
                const int pp = hf * PPH + p2;
                half4 ov;
                ov[0] = (_Float16)fmaf(fmaf(acc[s][pp][0], rstd[pp], nmr[pp]), gw.x, gb.x);
                ov[1] = (_Float16)fmaf(fmaf(acc[s][pp][1], rstd[pp], nmr[pp]), gw.y, gb.y);
                ov[2] = (_Float16)fmaf(fmaf(acc[s][pp][2], rstd[pp], nmr[pp]), gw.z, gb.z);
                ov[3] = (_Float16)fmaf(fmaf(acc[s][pp][3], rstd[pp], nmr[pp]), gw.w, gb.w);
                const int px = (2 * pp + rsel) * 16 + n;
                *reinterpret_cast<half4*>(out_s + px * (C * 2) + ((chunk ^ n) << 4) + (q & 1) * 8) = ov;
            }
        }
        GP_DWT_MARK(42 + 2 * hf);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // not __syncthreads(): the first half's stores stay in flight
        asm volatile("" ::: "memory");
        GP_DWT_MARK(43 + 2 * hf);
#pragma unroll 4
        for (int i = tid; i < (NPX / NHF) * cpp; i += 512) {
            const int px = hf * (NPX / NHF) + i / cpp, c = i % cpp;
            const uint4 v = *reinterpret_cast<const uint4*>(out_s + px * (C * 2) + ((c ^ (px & 15)) << 4));
            if constexpr (PAIR) *reinterpret_cast<uint4*>(yb + ((long)((px & 15) >> 3) * H * 8 + (px >> 4) * 8 + (px & 7)) * C + c * 8) = v;
            else *reinterpret_cast<uint4*>(yb + ((long)(px >> 4) * W + (px & 15)) * C + c * 8) = v;
        }
    }
    GP_DWT_MARK(46);
#ifdef GP_DW_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GP_DWT_MARK(47);
#endif
}

template <int NS, int J, bool WIDE = false, bool NOMFMA = false, int TH = 8, bool PAIR = false>
void launch_dw7_tall(const void* x, const void* wt, const float* bias, const float* lnw, const float* lnb, void* y, int B, int H, int W,
                     float eps, hipStream_t s) {
    constexpr int C = 64 * NS, NST = WIDE ? 1 : (NS >= 3 ? 3 : NS), STAGE = ((TH + 6) * (WIDE ? 22 : 18) * 128 + 1023) / 1024 * 1024 + 8 * 1024, OUTB = 16 * TH * C * 2;
    constexpr int PAR_OFF = (OUTB > NST * STAGE ? OUTB : NST * STAGE), LDS = PAR_OFF + ((3 * C * 4 + 1023) / 1024) * 1024 + 2 * 8 * 16 * TH * 4;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)dwconv7_ln_tall_kernel<NS, J, WIDE, NOMFMA, TH, PAIR>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((dwconv7_ln_tall_kernel<NS, J, WIDE, NOMFMA, TH, PAIR>), dim3(PAIR ? (B / 2) * (H / TH) : B * (H / TH) * (W / 16)), dim3(512), LDS, s, (const half_t*)x, (const half_t*)wt, bias,
                       lnw, lnb, (half_t*)y, H, W, eps);
}

// ---------------------------------------------------------------------------- row LayerNorm
// TI != T: fp32 input rows, fp16 output (dtype GP_F16 | GP_IN_F32: the fp32 residual stream of the fp16 mode enters the
// downsample LayerNorm in fp32); a thread then loads its 8 elements as two 16-byte vectors
template <typename T, typename TI = T>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, T* __restrict__ y,
                                                        long rows, int C, float eps, int ldy, long pl) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float red[4];
    const int CT = C / VEC, PG = 256 / CT;
    const int cs = threadIdx.x % CT;
    const long row = (long)blockIdx.x * PG + threadIdx.x / CT;
    const bool valid = row < rows;
    float a[VEC];
    if (valid) {
        if constexpr (std::is_same<T, TI>::value) {
            const Vec16<T> v = load16<T>(x + row * C + cs * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) a[e] = v.get(e);
        } else {
            static_assert(sizeof(TI) == 4 && VEC == 8, "mixed form: fp32 in, fp16 out");
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + row * C + cs * VEC), v1 = *reinterpret_cast<const f32x4*>(x + row * C + cs * VEC + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = v0[e]; a[e + 4] = v1[e]; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = 0.f;
    }
    float s[1] = {0.f};
#pragma unroll
    for (int e = 0; e < VEC; ++e) s[0] += a[e];
    pixel_group_sum<1>(s, CT, red);
    const float mean = s[0] / C;
    float q[1] = {0.f};
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        a[e] -= mean;
        q[0] += a[e] * a[e];
    }
    pixel_group_sum<1>(q, CT, red);
    if (!valid) return;
    const float rstd = rsqrtf(q[0] / C + eps);
    float gw[VEC], gb[VEC];
    load_f32<T>(w + cs * VEC, gw);
    load_f32<T>(b + cs * VEC, gb);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.set(e, a[e] * rstd * gw[e] + gb[e]);
    store16p<T>(y, row * ldy + cs * VEC, o, pl);
}

// GroupNorm scale / shift arithmetic.  fp16 storage: every rounding pinned (explicit fma), because gp_groupnorm_upsample2x must agree bit for
// bit with gp_groupnorm_apply and hipcc contracts `a * b + c` differently from one instantiation to the next.  fp32 storage keeps the
// plain expressions of rounds 1-3 (nothing has to agree with them, and the parity modes' measured errors stay what they were).
template <typename T> __device__ __forceinline__ float gn_shift(float mean, float sc, float gb) {
    if constexpr (sizeof(T) == 2) return __fmaf_rn(-mean, sc, gb);
    else return gb - mean * sc;
}
template <typename T> __device__ __forceinline__ float gn_norm(float v, float sc, float sh) {
    if constexpr (sizeof(T) == 2) return __fmaf_rn(v, sc, sh);
    else return v * sc + sh;
}
// ---------------------------------------------------------------------------- GroupNorm
// pixels per block: 256 when that still gives >= 1024 blocks, else 64 (small maps are latency bound)
static inline int gn_pxb(int B, int HW) { return ((long)B * HW / 256 >= 1024 || HW < 64) ? 256 : 64; }
// pixels per workgroup of the apply kernels (independent of the statistics chunks): small enough for >= 8 workgroups
// per CU at bs = 64, so that one workgroup's statistics finalize / VALU phase overlaps the others' streaming
static inline int gn_apply_pxb(int B, int HW) {
    int pxb = gn_pxb(B, HW);
    while (pxb > 32 && (long)B * cdiv(HW, pxb) < 2048) pxb >>= 1;
    return pxb;
}

// partial[((b*chunks + chunk)*G + g)*2 + {0,1}] = (sum, sum of squares) of this chunk, fixed order
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ x, float* __restrict__ partial,
                                                         int HW, int C, int G, int GN_PXB) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float part[256][4][2];
    const int CT = C / VEC, PG = 256 / CT, cpg = C / G;
    const int NG = cpg >= VEC ? 1 : VEC / cpg;  // groups covered by one 16-B vector (<= 4)
    const int cs = threadIdx.x % CT, pl = threadIdx.x / CT;
    const int b = blockIdx.y, chunk = blockIdx.x, chunks = gridDim.x;
    const int p0 = chunk * GN_PXB, p1 = min(HW, p0 + GN_PXB);
    float sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
    const T* xb = x + ((long)b * HW) * C + cs * VEC;
    for (int p = p0 + pl; p < p1; p += PG) {
        const Vec16<T> v = load16<T>(xb + (long)p * C);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float f = v.get(e);
            const int j = NG == 1 ? 0 : e / cpg;
            sm[j] += f;
            sq[j] += f * f;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        part[threadIdx.x][j][0] = sm[j];
        part[threadIdx.x][j][1] = sq[j];
    }
    __syncthreads();
    const int g = threadIdx.x;
    if (g < G) {
        float a = 0.f, q = 0.f;
        int cs0, cs1, j;
        if (NG == 1) {
            const int tpg = cpg / VEC;  // threads per group
            cs0 = g * tpg; cs1 = cs0 + tpg; j = 0;
        } else {
            cs0 = g / NG; cs1 = cs0 + 1; j = g % NG;
        }
        for (int l = 0; l < PG; ++l)
            for (int c = cs0; c < cs1; ++c) {
                a += part[l * CT + c][j][0];
                q += part[l * CT + c][j][1];
            }
        float* o = partial + (((long)b * chunks + chunk) * G + g) * 2;
        o[0] = a;
        o[1] = q;
    }
}

// (mean, rstd) of every group of image b from the chunk partials, into st[g][0..1]; ends with a barrier.
// All 256 threads take part: thread t sums the chunks c = t / G, t / G + 256 / G, ... of group t % G (fixed order),
// the 256 / G slices are added in fixed order through LDS.  The first version let thread g walk all chunks alone:
// 64 dependent L2 round trips (~40 us) in front of every 64x64 GroupNorm apply.
__device__ __forceinline__ void gn_finalize(const float* __restrict__ partial, int b, int chunks, int G, float inv_count,
                                            float eps, float (*st)[2]) {
    __shared__ double fin[256][2];
    const int g = threadIdx.x % G, sl = threadIdx.x / G, nsl = 256 / G;
    double a = 0.0, q = 0.0;
    for (int c = sl; c < chunks; c += nsl) {
        const f32x2 v = *reinterpret_cast<const f32x2*>(partial + (((long)b * chunks + c) * G + g) * 2);
        a += v[0];
        q += v[1];
    }
    fin[threadIdx.x][0] = a;
    fin[threadIdx.x][1] = q;
    __syncthreads();
    if (threadIdx.x < G) {
        a = 0.0; q = 0.0;
        for (int i = 0; i < nsl; ++i) {
            a += fin[i * G + threadIdx.x][0];
            q += fin[i * G + threadIdx.x][1];
        }
        const double mean = a * inv_count;
        double var = q * inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        st[threadIdx.x][0] = (float)mean;
        st[threadIdx.x][1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ x, const float* __restrict__ partial,
                                                       const float* __restrict__ w, const float* __restrict__ bb,
                                                       T* __restrict__ y, int HW, int C, int G, int act, int ldy,
                                                       int chunks, float inv_count, float eps, int GN_PXB, long plane) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float st[256][2];
    const int CT = C / VEC, PG = 256 / CT, cpg = C / G;
    const int cs = threadIdx.x % CT, pl = threadIdx.x / CT;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * GN_PXB, p1 = min(HW, p0 + GN_PXB);
    const T* xb = x + ((long)b * HW) * C + cs * VEC;
    const long yo = ((long)b * HW) * ldy + cs * VEC;
    // 4 pixels in flight per thread (the loop is otherwise one dependent HBM round trip per pixel); the first four
    // and the affine parameters are requested before the statistics are finalised, whose loads they then overlap
    Vec16<T> v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (p0 + pl + u * PG < p1) v[u] = load16<T>(xb + (long)(p0 + pl + u * PG) * C);
    float gw[VEC], gb[VEC];
    load_f32<T>(w + cs * VEC, gw);
    load_f32<T>(bb + cs * VEC, gb);
    gn_finalize(partial, b, chunks, G, inv_count, eps, st);
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int g = (cs * VEC + e) / cpg;
        sc[e] = st[g][1] * gw[e];
        sh[e] = gn_shift<T>(st[g][0], sc[e], gb[e]);
    }
    for (int pb = p0 + pl; pb < p1; pb += 4 * PG) {
        Vec16<T> cur[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cur[u] = v[u];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (pb + (4 + u) * PG < p1) v[u] = load16<T>(xb + (long)(pb + (4 + u) * PG) * C);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = pb + u * PG;
            if (p >= p1) break;
            Vec16<T> o;
            if (act == GP_ACT_GELU && sizeof(T) == 2) {
#pragma unroll
                for (int e = 0; e < VEC; e += 2) {
                    const f32x2 g = gelu_poly2(f32x2{gn_norm<T>(cur[u].get(e), sc[e], sh[e]), gn_norm<T>(cur[u].get(e + 1), sc[e + 1], sh[e + 1])});
                    o.set(e, g[0]);
                    o.set(e + 1, g[1]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) o.set(e, apply_act(gn_norm<T>(cur[u].get(e), sc[e], sh[e]), act));
            }
            store16p<T>(y, yo + (long)p * ldy, o, plane);
        }
    }
}

// GroupNorm apply + activation + the head's 1x1 out layer (C -> 3) in one pass: the normalised 64x64x256 tensor is
// consumed only by that out layer (network/xyz_head.py:352-357), so it is never written -- saves a 134 MB write
// and a 134 MB read per head at bs = 64.  out_nchw (B,3,HW) fp32 + out_nhwc4 (B*HW,4) fp32, as gp_xyz_out_layer.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_xyz_kernel(const T* __restrict__ x, const float* __restrict__ partial,
                                                           const float* __restrict__ w, const float* __restrict__ bb,
                                                           const float* __restrict__ ow, const float* __restrict__ ob,
                                                           float* __restrict__ out_nchw, float* __restrict__ out_nhwc4,
                                                           int HW, int C, int G, int act, int chunks, float inv_count,
                                                           float eps, int GN_PXB) {
    constexpr int VEC = Vec16<T>::N;
    __shared__ float st[256][2];
    __shared__ float red[4 * 3];
    const int CT = C / VEC, PG = 256 / CT, cpg = C / G;
    const int cs = threadIdx.x % CT, pl = threadIdx.x / CT;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * GN_PXB, p1 = min(HW, p0 + GN_PXB);
    const T* xb = x + ((long)b * HW) * C + cs * VEC;
    const int iters = (GN_PXB + PG - 1) / PG;   // uniform trip count: the reduction below needs every thread
    Vec16<T> vn[4];                             // 4 pixels in flight per thread, statically indexed; first four
#pragma unroll                                  // requested ahead of the statistics finalize
    for (int u = 0; u < 4; ++u) {
        const int p = p0 + pl + u * PG;
        if (u < iters && p < p1) vn[u] = load16<T>(xb + (long)p * C);
    }
    float sc[VEC], sh[VEC], w0[VEC], w1[VEC], w2[VEC];
    float gw[VEC], gb[VEC];
    load_f32<T>(w + cs * VEC, gw);
    load_f32<T>(bb + cs * VEC, gb);
    load_f32<T>(ow + cs * VEC, w0);
    load_f32<T>(ow + C + cs * VEC, w1);
    load_f32<T>(ow + 2 * C + cs * VEC, w2);
    const float b0 = ob[0], b1 = ob[1], b2 = ob[2];
    gn_finalize(partial, b, chunks, G, inv_count, eps, st);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int g = (cs * VEC + e) / cpg;
        sc[e] = st[g][1] * gw[e];
        sh[e] = gn_shift<T>(st[g][0], sc[e], gb[e]);
    }
    for (int it0 = 0; it0 < iters; it0 += 4) {
        Vec16<T> vq[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vq[u] = vn[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = p0 + pl + (it0 + 4 + u) * PG;
            if (it0 + 4 + u < iters && p < p1) vn[u] = load16<T>(xb + (long)p * C);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (it0 + u >= iters) break;                  // uniform
            const int p = p0 + pl + (it0 + u) * PG;
            float d[3] = {0.f, 0.f, 0.f};
            if (p < p1) {
                const Vec16<T> v = vq[u];
                float a[VEC];
                if (act == GP_ACT_GELU && sizeof(T) == 2) {
#pragma unroll
                    for (int e = 0; e < VEC; e += 2) {
                        const f32x2 g = gelu_poly2(f32x2{gn_norm<T>(v.get(e), sc[e], sh[e]), gn_norm<T>(v.get(e + 1), sc[e + 1], sh[e + 1])});
                        a[e] = g[0];
                        a[e + 1] = g[1];
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) a[e] = apply_act(gn_norm<T>(v.get(e), sc[e], sh[e]), act);
                }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    d[0] = fmaf(a[e], w0[e], d[0]);
                    d[1] = fmaf(a[e], w1[e], d[1]);
                    d[2] = fmaf(a[e], w2[e], d[2]);
                }
            }
            pixel_group_sum<3>(d, CT, red);
            if (p < p1 && cs == 0) {
                const long row = (long)b * HW + p;
                out_nchw[((long)b * 3 + 0) * HW + p] = d[0] + b0;
                out_nchw[((long)b * 3 + 1) * HW + p] = d[1] + b1;
                out_nchw[((long)b * 3 + 2) * HW + p] = d[2] + b2;
                *reinterpret_cast<f32x4*>(out_nhwc4 + row * 4) = f32x4{d[0] + b0, d[1] + b1, d[2] + b2, 0.f};
            }
        }
    }
}

// The same pass with the 1x1 out layer on the matrix pipe (fp16 input, C = 256, GELU): per 16 pixels a wave normalises and
// activates its pixels' channels in the MFMA B layout (lane (pixel m, fq) owns channels ks*32 + fq*8 .. +8 of K step ks: one
// 16-byte load each), splits them into fp16 hi + lo, and 8 x 3 MFMAs against the (3 -> 16 rows, fp16 hi + lo) out-layer weights
// do the 3 x 256 dot products and their reduction in fp32 accuracy: no per-pixel cross-lane sums, no fp32 dot FMAs.
// scale / shift per channel in LDS.
// H16 (round 5; act | GP_ACT_PACKED16: PoseNet's fp16 mode asks for it, PoseNetConfig.gnxyz16 / GP_GNXYZ16=0 switch back): the affine and the GELU on PACKED fp16 -- v_pk_fma_f16 on the stored fp16 values against (scale, shift)
// pairs rounded to fp16, then gelu16_slice (common.hpp) -- and the packed results go into the MFMA as they stand (hi / lo weight fragments as before, no lo activations):
// 13 VALU operations per value PAIR instead of ~34 (2 conversions + 2 FMAs + 2 x 12 polynomial + 6 for the hi / lo split): the fp32 form is VALU-bound (81 us per
// 128 crops at 3.5 TB/s where the same pass without an out layer streams at 5.7), profiles/r05_gnxyz16_ab.txt.
template <bool H16>
__global__ __launch_bounds__(256, 4) void gn_apply_xyz_mfma_kernel(const half_t* __restrict__ x, const float* __restrict__ partial,
                                                                const float* __restrict__ w, const float* __restrict__ bb,
                                                                const float* __restrict__ ow, const float* __restrict__ ob,
                                                                float* __restrict__ out_nchw, float* __restrict__ out_nhwc4,
                                                                int HW, int G, int chunks, float inv_count, float eps, int GN_PXB) {
    constexpr int C = 256, KS = C / 32;
    __shared__ float st[256][2];
    __shared__ __attribute__((aligned(16))) float sc_s[C], sh_s[C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * GN_PXB, p1 = min(HW, p0 + GN_PXB);
    // out-layer weights as A fragments in LDS ([ks][hi / lo][lane] x 16 B; 64 registers otherwise): row n = fr (3 real rows),
    // k = ks*32 + fq*8 + j; fp16 hi + lo of the fp32 weights.  Wave w builds K steps 2w, 2w + 1.
    __shared__ __attribute__((aligned(16))) half8 a_s[KS][2][64];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int ks = wave * 2 + kk;
        half8 hi8, lo8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = fr < 3 ? ow[fr * C + ks * 32 + fq * 8 + j] : 0.f;
            const half_t hi = (half_t)v;
            hi8[j] = hi;
            lo8[j] = (half_t)(v - (float)hi);
        }
        a_s[ks][0][lane] = hi8;
        a_s[ks][1][lane] = lo8;
    }
    gn_finalize(partial, b, chunks, G, inv_count, eps, st);
    __shared__ __attribute__((aligned(16))) half_t sc16_s[C], sh16_s[C];
    {
        const int cpg = C / G, g = tid / cpg;
        const float s = st[g][1] * w[tid];
        sc_s[tid] = s;
        sh_s[tid] = bb[tid] - st[g][0] * s;
        sc16_s[tid] = (half_t)s;
        sh16_s[tid] = (half_t)(bb[tid] - st[g][0] * s);
    }
    __syncthreads();
    const float b0 = ob[0], b1 = ob[1], b2 = ob[2];
    for (int pb = p0 + wave * 16; pb < p1; pb += 64) {
        const int p = pb + fr;
        const bool ok = p < p1;
        const half_t* xp = x + ((long)b * HW + (ok ? p : p0)) * C + fq * 8;
        half8 xv[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xv[ks] = *reinterpret_cast<const half8*>(xp + ks * 32);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = ks * 32 + fq * 8;
            if constexpr (H16) {
                const half8 s8 = *reinterpret_cast<const half8*>(sc16_s + c0), h8 = *reinterpret_cast<const half8*>(sh16_s + c0);
                half2v hx[4], hu[4], hp[4];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    hx[i] = __builtin_elementwise_fma(half2v{xv[ks][2 * i], xv[ks][2 * i + 1]}, half2v{s8[2 * i], s8[2 * i + 1]}, half2v{h8[2 * i], h8[2 * i + 1]});
                static_for<1, GELU16_SLICES>([&](auto sc) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gelu16_slice<decltype(sc)::value>(f32x2{0.f, 0.f}, hx[i], hu[i], hp[i]);
                });
                const half8 a16 = half8{hp[0][0], hp[0][1], hp[1][0], hp[1][1], hp[2][0], hp[2][1], hp[3][0], hp[3][1]};
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_s[ks][1][lane], a16, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_s[ks][0][lane], a16, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc_s + c0), s1 = *reinterpret_cast<const f32x4*>(sc_s + c0 + 4);
            const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh_s + c0), h1 = *reinterpret_cast<const f32x4*>(sh_s + c0 + 4);
            f32x2 v[4];
            v[0] = f32x2{fmaf((float)xv[ks][0], s0[0], h0[0]), fmaf((float)xv[ks][1], s0[1], h0[1])};
            v[1] = f32x2{fmaf((float)xv[ks][2], s0[2], h0[2]), fmaf((float)xv[ks][3], s0[3], h0[3])};
            v[2] = f32x2{fmaf((float)xv[ks][4], s1[0], h1[0]), fmaf((float)xv[ks][5], s1[1], h1[1])};
            v[3] = f32x2{fmaf((float)xv[ks][6], s1[2], h1[2]), fmaf((float)xv[ks][7], s1[3], h1[3])};
            {   // GELU as 12 slices of one inline-asm v_fma_f32 / v_mul_f32 per element (common.hpp): a fixed instruction sequence
                f32x2 gx[4], gt[4], gp[4];
                float c1v = GELU_H[1];
                asm volatile("" : "+v"(c1v));
                static_for<0, GELU_SLICES>([&](auto sc) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) gelu_poly2_slice<decltype(sc)::value>(v[i], gx[i], gt[i], gp[i], c1v);
                });
            }
            half8 a16, a16l;   // activations as fp16 hi + lo as well: the fused pass keeps the fp32 accuracy of the VALU form
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const half_t hi = (half_t)v[i][e];
                    a16[2 * i + e] = hi;
                    a16l[2 * i + e] = (half_t)(v[i][e] - (float)hi);
                }
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_s[ks][1][lane], a16, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_s[ks][0][lane], a16l, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a_s[ks][0][lane], a16, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);   // one K step's four GELU chains at a time (all eight in flight: 254 registers)
        }
        // D: lane (pixel fr, fq) holds out rows fq*4 + e: the three outputs live in the fq == 0 lanes
        if (fq == 0 && ok) {
            const long row = (long)b * HW + p;
            const float d0 = acc[0] + b0, d1 = acc[1] + b1, d2 = acc[2] + b2;
            out_nchw[((long)b * 3 + 0) * HW + p] = d0;
            out_nchw[((long)b * 3 + 1) * HW + p] = d1;
            out_nchw[((long)b * 3 + 2) * HW + p] = d2;
            *reinterpret_cast<f32x4*>(out_nhwc4 + row * 4) = f32x4{d0, d1, d2, 0.f};
        }
    }
}

bool ct_ok(int C, int esz) {
    const int vec = 16 / esz;
    if (C % vec) return false;
    const int ct = C / vec;
    return ct >= 1 && ct <= 256 && (ct & (ct - 1)) == 0;
}


// ---------------------------------------------------------------------------- GroupNorm apply + GELU + bilinear x2 in one pass (round 4)
// TopDownXyzHead runs conv -> GN -> GELU -> Upsample(x2, bilinear, align_corners) -> conv twice (xyz_head.py:250-264); until round 4
// that was a GroupNorm-apply pass (read + write of the low-resolution tensor) and an upsample pass.  Here a workgroup owns a 16 x 8
// OUTPUT tile of one image: its 10 x 6 source pixels are normalised + activated ONCE into LDS (rounded to fp16 exactly as the
// apply pass stored them), then the output pixels are blended from LDS with the upsample kernel's own roundings (common.hpp: bilerp_*)
// -- bitwise the two-pass result (tests/test_hip_ops.py), one launch and one low-resolution round trip less.  (Fusing the GELU into
// the per-output-pixel upsample kernel instead would run it four times per source value: 230 us of VALU issue at 128 crops.)
template <int CPT>   // output columns per thread in the blend: 16 / (256 / (C / 8)), at least 1
__global__ __launch_bounds__(256) void gn_upsample2x_kernel(const half_t* __restrict__ x, const float* __restrict__ partial,
                                                            const float* __restrict__ w, const float* __restrict__ bb,
                                                            half_t* __restrict__ y, int H, int W, int C, int G, int act,
                                                            int chunks, float inv_count, float eps) {
    // 16 x 8 output tile <- 10 x 6 source pixels (30 KB of LDS at C = 256: 4 workgroups per CU).  16 x 16 tiles (less halo, half
    // the workgroups per CU) measured the same at 128 crops and lose at B = 1: profiles/r04_gn_upsample_ab.txt.
    constexpr int TO = 16, TS = TO / 2 + 2, TOY = 8, TSY = TOY / 2 + 2;
    extern __shared__ __attribute__((aligned(16))) char gus[];
    __shared__ float st[256][2];
    const int CT = C >> 3, PG = 256 / CT, cpg = C / G;
    const int cs = threadIdx.x % CT, pl = threadIdx.x / CT;
    const int Ho = 2 * H, Wo = 2 * W;
    const int b = blockIdx.z, oy0 = blockIdx.y * TOY, ox0 = blockIdx.x * TO;
    // align_corners = True: src = dst * (in - 1) / (out - 1); the tile's first source row / column
    const float ry = (float)(H - 1) / (float)(Ho - 1), rx = (float)(W - 1) / (float)(Wo - 1);
    const int sy0 = (int)bilerp_src(ry, oy0), sx0 = (int)bilerp_src(rx, ox0);
    float gw[8], gb[8];
    load_f32<half_t>(w + cs * 8, gw);
    load_f32<half_t>(bb + cs * 8, gb);
    gn_finalize(partial, b, chunks, G, inv_count, eps, st);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (cs * 8 + e) / cpg;
        sc[e] = st[g][1] * gw[e];
        sh[e] = gn_shift<half_t>(st[g][0], sc[e], gb[e]);
    }
    const half_t* xb = x + ((long)b * H * W) * C + cs * 8;
    const int NSRC = TSY * TS;
    for (int pb = pl; pb < NSRC; pb += 4 * PG) {      // phase 1: source tile -> GN + act -> LDS (fp16); four loads in flight per thread
        Vec16<half_t> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = min(pb + u * PG, NSRC - 1);
            const int ty = p / TS, tx = p - ty * TS;
            const int gy = min(sy0 + ty, H - 1), gx = min(sx0 + tx, W - 1);
            v[u] = load16<half_t>(xb + ((long)gy * W + gx) * C);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = pb + u * PG;
            if (p >= NSRC) break;
            Vec16<half_t> o;
            if (act == GP_ACT_GELU) {
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    const f32x2 g = gelu_poly2(f32x2{gn_norm<half_t>(v[u].get(e), sc[e], sh[e]), gn_norm<half_t>(v[u].get(e + 1), sc[e + 1], sh[e + 1])});
                    o.set(e, g[0]);
                    o.set(e + 1, g[1]);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) o.set(e, apply_act(gn_norm<half_t>(v[u].get(e), sc[e], sh[e]), act));
            }
            *reinterpret_cast<uint4*>(gus + ((long)p * CT + cs) * 16) = o.u;
        }
    }
    __syncthreads();
    // phase 2: the blend is separable -- h(row) = hx * v[row][x0] + lx * v[row][x1], out = hy * h(y0) + ly * h(y1), with exactly the
    // upsample kernel's roundings (misc.hip: bilerp) -- and a source row's h serves every output row that touches it, so it is
    // computed once and rolled (12 -> 5 VALU operations per output value).  A thread owns CPT adjacent output columns of one
    // 8-channel slice and walks the tile's output rows in order; all row decisions are wave-uniform.
    if (pl * CPT >= TO) return;                       // C < 128: more pixel groups than columns (after the last barrier)
    float h0[CPT][8], h1[CPT][8], lxs[CPT];
    int xo[CPT];
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
        const int ox = min(ox0 + pl * CPT + j, Wo - 1);
        const float sx = bilerp_src(rx, ox);
        const int x0 = (int)sx;
        lxs[j] = sx - x0;
        xo[j] = ((x0 - sx0) * CT + cs) * 16;
    }
    auto hrow = [&](int row, float (&h)[CPT][8]) {
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            Vec16<half_t> va, vb;
            va.u = *reinterpret_cast<const uint4*>(gus + (long)row * TS * CT * 16 + xo[j]);
            vb.u = *reinterpret_cast<const uint4*>(gus + (long)row * TS * CT * 16 + xo[j] + CT * 16);
            const float lx = lxs[j], hx = 1.f - lx;
#pragma unroll
            for (int e = 0; e < 8; ++e) h[j][e] = bilerp_h(hx, va.get(e), lx, vb.get(e));
        }
    };
    int have = -2;
    const int oyend = min(oy0 + TOY, Ho);
    for (int oy = oy0; oy < oyend; ++oy) {
        const float sy = bilerp_src(ry, oy);
        const int y0 = (int)sy, r = y0 - sy0;
        const float ly = sy - y0, hy = 1.f - ly;
        if (r != have) {                              // uniform
            if (r == have + 1) {
#pragma unroll
                for (int j = 0; j < CPT; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) h0[j][e] = h1[j][e];
            } else {
                hrow(r, h0);
            }
            hrow(r + 1, h1);                          // LDS row r + 1 holds source row min(y0 + 1, H - 1)
            have = r;
        }
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            const int ox = ox0 + pl * CPT + j;
            if (ox >= Wo) break;
            Vec16<half_t> o;
            bilerp_v_vec<half_t>(o, hy, h0[j], ly, h1[j]);
            store16<half_t>(y + (((long)b * Ho + oy) * Wo + ox) * C + cs * 8, o);
        }
    }
}

}  // namespace

static int dw3_tile_rows() {   // GP_DW3_TILE_ROWS=4|2: rows of a dwconv3_ln_tile_kernel tile (A/B)
    static const int k = [] { const char* e = getenv("GP_DW3_TILE_ROWS"); return e && atoi(e) == 4 ? 4 : 2; }();
    return k;
}
static long dw3_tile_min() {   // fewest 16 x 4 tiles that go to dwconv3_ln_tile_kernel (GP_DW3_TILE_MIN: A/B; a huge value switches it off)
    static const long k = [] { const char* e = getenv("GP_DW3_TILE_MIN"); return e ? atol(e) : 256l; }();
    return k;
}
static bool dw3_narrow_always() {   // GP_DW3_NARROW=0: the 3 x 3 strip kernel back on 8 pixels per thread above dw_narrow_below() workgroups (A/B)
    static const bool on = [] { const char* e = getenv("GP_DW3_NARROW"); return !(e && e[0] == '0'); }();
    return on;
}
static long dw_narrow_below() {   // strip kernel: 2 pixels per thread when 8 would give fewer workgroups than this (GP_DW_NARROW_BELOW: A/B)
    static const long k = [] { const char* e = getenv("GP_DW_NARROW_BELOW"); return e ? atol(e) : 128l; }();
    return k;
}
// fewest workgroups (4 x 16-pixel tiles) that still go to the MFMA kernel: its workgroups live ~16 us whatever the batch (weight
// fragments, halo pipeline), the strip kernel at 2 pixels per thread ~9.5 us while it fits the chip (scripts/dw_small_ab.py, hipGraph
// chains: C = 512 15.9 vs 9.5 us at 1-4 crops; C = 256 10.4 vs 9.5 at 1 crop, 10.6 vs 11.5 at 4; C = 128: MFMA always.  Inside the
// network (scripts/b1_trace.py) the strip kernel wins up to 12 crops at C = 512 -- 3.25 vs 3.37 ms -- and loses at 16: 4.04 vs 3.62).  GP_DW_MFMA_MIN=<n>: one threshold for every C (A/B).
static long dw_mfma_min_wgs(int C) {
    static const long k = [] { const char* e = getenv("GP_DW_MFMA_MIN"); return e ? atol(e) : -1l; }();
    return k >= 0 ? k : C == 512 ? 52 : C == 256 ? 33 : 0;
}
// fewest half-image workgroups that go to dwconv7_ln_tall_kernel on 16-wide maps (one workgroup per CU: at 64 crops = 128 workgroups the 16 x 4 tiles are level,
// 19.5 against 20.2 us; at 80 / 96 crops it is 33.5 against 21 us: profiles/r06_dw_tall_ab.txt);
// GP_DW_TALL_MIN=<n>: A/B (a huge value switches the kernel off)
static long dw_tall_min_wgs() {
    static const long k = [] { const char* e = getenv("GP_DW_TALL_MIN"); return e ? atol(e) : 130l; }();
    return k;
}
// the column-halo form on maps wider than 16 pixels, from this many 16 x 8 tiles up (profiles/r06_dw_tall_ab.txt: C = 128 at 64 x 64 wins from 16 crops,
// C = 256 at 32 x 32 from 32; GP_DW_TALLW_MIN: one threshold for both, A/B; a huge value switches it off)
static long dw_tallw_min_wgs(int C) {
    static const long k = [] { const char* e = getenv("GP_DW_TALLW_MIN"); return e ? atol(e) : -1l; }();
    return k >= 0 ? k : C == 128 ? 384 : 256;
}
static int dw_pair_rows() {   // GP_DW_PAIR_ROWS=4|2: tile height of the pair form (A/B)
    static const int k = [] { const char* e = getenv("GP_DW_PAIR_ROWS"); return e && atoi(e) == 4 ? 4 : 2; }();
    return k;
}
static long dw_pair_min_crops() {     // the pair-tile form at stage 3 (GP_DW_PAIR_MIN: A/B; a huge value switches it off)
    static const long k = [] { const char* e = getenv("GP_DW_PAIR_MIN"); return e ? atol(e) : 32l; }();
    return k;
}
static bool dw_tall4_enabled() {     // GP_DW_TALL4=0: A/B switch for the quarter-image form
    static const bool on = [] { const char* e = getenv("GP_DW_TALL4"); return !(e && e[0] == '0'); }();
    return on;
}
static bool dw_single_buffer() {   // A/B: the single-buffered (two workgroups per CU) C = 512 variant for any grid
    static const bool on = [] { const char* e = getenv("GP_DW_NBUF1"); return e && e[0] == '1'; }();
    return on;
}

extern "C" int gp_dwconv_ln(const void* x, const void* wt, const float* bias, const float* ln_w,
                            const float* ln_b, void* y, int B, int H, int W, int C, int KS, float eps, int act,
                            long n_pixels, int dtype_in, void* stream) {
    GP_REQUIRE(x && wt && bias && ln_w && ln_b && y, "gp_dwconv_ln: null pointer");
    const int dtype = dtype_in & ~GP_OUT_PLANES;
    GP_REQUIRE(dtype == GP_F32 || dtype == GP_F16, "gp_dwconv_ln: bad dtype");
    GP_REQUIRE(!(dtype_in & GP_OUT_PLANES) || (dtype == GP_F32 && x != y), "gp_dwconv_ln: GP_OUT_PLANES needs GP_F32 and y != x");
    const long pl = (dtype_in & GP_OUT_PLANES) ? (long)B * H * W * C : 0;    // lo' plane: one dense tensor behind hi
    const int esz = dtype == GP_F16 ? 2 : 4;
    GP_REQUIRE(ct_ok(C, esz) && C / (16 / esz) >= 16, "gp_dwconv_ln: unsupported C=%d", C);
    GP_REQUIRE(KS == 3 || KS == 7, "gp_dwconv_ln: KS=%d unsupported (3 or 7)", KS);
    GP_REQUIRE(W % 8 == 0, "gp_dwconv_ln: W=%d must be a multiple of 8", W);
    const long total = (long)B * H * W;
    GP_REQUIRE(n_pixels > 0 && n_pixels <= total, "gp_dwconv_ln: n_pixels out of range");
    const int CT = C / (16 / esz), PG = 256 / CT;
    const long strips = (n_pixels + 7) / 8;
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_DWCONV_LN, 2.0 * n_pixels * C * KS * KS, (double)n_pixels * C * esz * 2);
    gp_timing_label("dwconv%d_ln C%d %dx%d B%d", KS, C, H, W, B);
    const int dbg = act >= 100 ? act - 100 : 0;   // 101 / 102: timing-only ablations (no conv / no DMA), wrong results
    if (act >= 100) act = GP_ACT_NONE;
    // 16 x 8 tiles (dwconv7_ln_tall_kernel): 16-pixel-wide maps (ConvNeXt stage 2) from dw_tall_min_wgs() workgroups up, wider maps (stages 0 / 1: the
    // column-halo form) from dw_tallw_min_wgs() up; act code 110 forces it, 111 (investigation builds): its no-MFMA ablation
    const bool tall16 = W == 16 && (C == 128 || C == 256 || C == 512), tallw = W > 16 && W % 16 == 0 && (C == 128 || C == 256);
    if (KS == 7 && act == GP_ACT_NONE && n_pixels == total && dtype == GP_F16 && H % 8 == 0 && (tall16 || tallw) && x != y &&
        (dbg == 10 || (dbg == 11 && C == 512 && H <= 16) || (dbg == 0 && (long)B * (H / 8) * (W / 16) >= (tall16 ? dw_tall_min_wgs() : dw_tallw_min_wgs(C))))) {
#ifdef GP_DW_STAMPS      // investigation builds only (the instantiation spills): act code 111 = no MFMAs, timing only
        if (dbg == 11) launch_dw7_tall<8, 8, false, true>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
        else
#endif
        if (tallw) {
            if (C == 128) launch_dw7_tall<2, 6, true>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
            else launch_dw7_tall<4, 6, true>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
        } else if (H <= 16) {
            if (C == 128) launch_dw7_tall<2, 8>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
            else if (C == 256) launch_dw7_tall<4, 8>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
            else launch_dw7_tall<8, 8>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
        } else {
            if (C == 128) launch_dw7_tall<2, 9>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
            else if (C == 256) launch_dw7_tall<4, 9>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
            else launch_dw7_tall<8, 9>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
        }
        GP_LAUNCH_CHECK("gp_dwconv_ln");
    }
    // 8 x 8 maps at C = 1024 (ConvNeXt stage 3): two images per 16-column tile, from dw_pair_min_crops() crops up (even batch); act code 113 forces it
    if (KS == 7 && act == GP_ACT_NONE && n_pixels == total && dtype == GP_F16 && W == 8 && H % 4 == 0 && C == 1024 && B % 2 == 0 && x != y &&
        (dbg == 13 || dbg == 14 || (dbg == 0 && B >= dw_pair_min_crops()))) {
        // quarter-image tiles of the pair (TH = 2, the default: 2 B workgroups, two B fragments per two MFMAs: 23.3 -> 18.7 us per 128 crops, 21.8 -> 16.7 per 64; act code 114)
        // or half-image tiles (TH = 4: act code 113, GP_DW_PAIR_ROWS=4); same bits
        if (dbg == 14 || (dbg == 0 && dw_pair_rows() == 2)) launch_dw7_tall<16, 8, false, false, 2, true>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
        else launch_dw7_tall<16, 8, false, false, 4, true>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
        GP_LAUNCH_CHECK("gp_dwconv_ln");
    }
    // quarter-image tiles (TH = 4) of 16-wide maps where the half-image tiles would leave the chip half empty (33 .. 64 crops at stage 2); act code 112 forces it
    if (KS == 7 && act == GP_ACT_NONE && n_pixels == total && dtype == GP_F16 && H % 4 == 0 && tall16 && x != y &&
        (dbg == 12 || (dbg == 0 && (long)B * (H / 4) >= 52 && (long)B * (H / 8) < dw_tall_min_wgs() && dw_tall4_enabled()))) {      // (from 13 crops: where the 16 x 4 MFMA kernel took over from the strip kernel; 16 crops: 12.5 against 16.4 us)
        if (C == 128) launch_dw7_tall<2, 8, false, false, 4>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
        else if (C == 256) launch_dw7_tall<4, 8, false, false, 4>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
        else launch_dw7_tall<8, 8, false, false, 4>(x, wt, bias, ln_w, ln_b, y, B, H, W, eps, s);
        GP_LAUNCH_CHECK("gp_dwconv_ln");
    }
    if (KS == 7 && act == GP_ACT_NONE && n_pixels == total && dtype == GP_F16 && (dbg == 0 || dbg >= 5) && dbg < 10 && H % 4 == 0 && W % 16 == 0 &&
        (C == 128 || C == 256 || C == 512) && (long)B * (H / 4) * (W / 16) >= dw_mfma_min_wgs(C)) {
        if (C == 128) launch_dw7_mfma<1, 1>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg);
        else if (C == 256) launch_dw7_mfma<2, 1>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg);
        else if ((long)B * (H / 4) * (W / 16) >= 512 || dw_single_buffer()) launch_dw7_mfma<4, 1>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg);
        else launch_dw7_mfma<4, 2>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg);
        GP_LAUNCH_CHECK("gp_dwconv_ln");
    }
    // the LDS-tiled VALU kernel needs >= ~192 tiles of 8x8 pixels to fill the chip; below that (stage 3 at bs = 64: one
    // tile per image, 64 workgroups) the strip kernel with 16 pixels per workgroup is 1.7x faster (scripts/dw_bench.py)
    if (KS == 7 && act == GP_ACT_NONE && n_pixels == total && H % 8 == 0 && W % 8 == 0 && dbg != 7 &&
        ((long)B * (H / 8) * (W / 8) >= 192 || dbg == 4)) {   // act code 104 forces it (tests)
        const int nslab = C / (16 * (16 / esz));
        bool done = true;
        if (dtype == GP_F16) {
            if (nslab == 1) launch_dw7_tiled<half_t, 1>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg);
            else if (nslab == 2) launch_dw7_tiled<half_t, 2>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg);
            else if (nslab == 4) launch_dw7_tiled<half_t, 4>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg);
            else done = false;   // C = 1024 in fp16: the 8-slab instantiation spilled to scratch (banned) -> strip kernel
        } else {
            if (nslab == 2) launch_dw7_tiled<float, 2>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg, pl);
            else if (nslab == 4) launch_dw7_tiled<float, 4>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg, pl);
            else if (nslab == 8) launch_dw7_tiled<float, 8>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg, pl);
            else if (nslab == 16) launch_dw7_tiled<float, 16>(x, wt, bias, ln_w, ln_b, y, B, H, W, C, eps, s, dbg, pl);
            else done = false;
        }
        if (done) GP_LAUNCH_CHECK("gp_dwconv_ln");
    }
    // KS = 3 (the DCNv3 prefix kernel), fp16, C = 256, whole 4-row blocks of 16-pixel-wide column tiles: the LDS-tiled kernel from dw3_tile_min() tiles up
    // (act code 120 + act forces it: tests)
    {
        const int dbg3 = dbg >= 20 && dbg < 30 ? 1 : 0;      // (act codes >= 100 were turned into dbg = code - 100 above)
        const int act3 = dbg3 ? dbg - 20 : act;
        const int th3 = dbg3 ? (dbg >= 25 ? 2 : 4) : dw3_tile_rows();      // (act codes 120 + act: the 4-row tile, 125 + act: the 2-row tile)
        const int act3b = dbg3 && dbg >= 25 ? dbg - 25 : act3;
        if (KS == 3 && dtype == GP_F16 && C == 256 && pl == 0 && act3b == GP_ACT_GELU && H % th3 == 0 && W % 16 == 0 && n_pixels % ((long)th3 * W) == 0 &&
            x != y && (dbg3 || n_pixels / 64 >= dw3_tile_min())) {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)dwconv3_ln_tile_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 18 * 512);
                (void)hipFuncSetAttribute((const void*)dwconv3_ln_tile_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 18 * 512);
                attr_set = true;
            }
            if (th3 == 4) hipLaunchKernelGGL(dwconv3_ln_tile_kernel<4>, dim3((unsigned)(n_pixels / 64)), dim3(256), 6 * 18 * 512, s, (const half_t*)x, (const half_t*)wt, bias, ln_w, ln_b, (half_t*)y, H, W, eps);
            else hipLaunchKernelGGL(dwconv3_ln_tile_kernel<2>, dim3((unsigned)(n_pixels / 32)), dim3(256), 4 * 18 * 512, s, (const half_t*)x, (const half_t*)wt, bias, ln_w, ln_b, (half_t*)y, H, W, eps);
            GP_LAUNCH_CHECK("gp_dwconv_ln");
        }
        GP_REQUIRE(!dbg3, "gp_dwconv_ln: act code 12x (forced 3 x 3 tile kernel) on a shape it does not take");
    }
    // few pixels (fewer than one 8-pixel-strip workgroup per two CUs): 2 pixels per thread, four times the workgroups
    // (KS = 3, the DCNv3 prefix kernel: the 2-pixel form with its rolling prefetch wins at every size -- 64 x 64 / 32 x 32 prefix of 64 crops 36.6 / 15.2 -> 28.9 / 10.1 us,
    // 128 crops 62 -> 53: profiles/r06_dw3_forms.txt; GP_DW3_NARROW=0: A/B switch)
    const bool narrow = dtype == GP_F16 && (cdiv(strips, PG) < dw_narrow_below() || (KS == 3 && dw3_narrow_always()));
    dim3 grid(narrow ? cdiv((n_pixels + 1) / 2, PG) : cdiv(strips, PG));
#define GP_DW(T, K, P) hipLaunchKernelGGL((dwconv_ln_kernel<T, K, P>), grid, dim3(256), 0, s, (const T*)x, (const T*)wt, bias, ln_w, ln_b, (T*)y, H, W, C, eps, act, n_pixels, pl)
    if (dtype == GP_F16) {
        if (narrow) { if (KS == 7) GP_DW(half_t, 7, 2); else GP_DW(half_t, 3, 2); }
        else { if (KS == 7) GP_DW(half_t, 7, 8); else GP_DW(half_t, 3, 8); }
    }
    else { if (KS == 7) GP_DW(float, 7, 8); else GP_DW(float, 3, 8); }
#undef GP_DW
    GP_LAUNCH_CHECK("gp_dwconv_ln");
}

extern "C" int gp_dwconv_ln_groups(const void* x, const void* wt, const float* bias, const float* ln_w, const float* ln_b, void* y, int B,
                                   int H, int W, int C, int KS, float eps, int act, const int* crop_group_start, int dtype, void* stream) {
    GP_REQUIRE(x && wt && bias && ln_w && ln_b && y && crop_group_start, "gp_dwconv_ln_groups: null pointer");
    GP_REQUIRE(dtype == GP_F32 || dtype == GP_F16, "gp_dwconv_ln_groups: bad dtype");
    const int esz = dtype == GP_F16 ? 2 : 4;
    GP_REQUIRE(ct_ok(C, esz) && C / (16 / esz) >= 16, "gp_dwconv_ln_groups: unsupported C=%d", C);
    GP_REQUIRE(KS == 3 || KS == 7, "gp_dwconv_ln_groups: KS=%d unsupported (3 or 7)", KS);
    GP_REQUIRE(B > 0 && W % 16 == 0 && H % 2 == 0 && x != y, "gp_dwconv_ln_groups: B=%d, W=%d %% 16, H=%d %% 2, y != x required", B, W, H);
    GP_REQUIRE(act == GP_ACT_NONE || act == GP_ACT_GELU || act == GP_ACT_RELU || act == GP_ACT_LRELU, "gp_dwconv_ln_groups: bad act");
    const long n_pixels = (long)B * H * W / 4;
    const int CT = C / (16 / esz), PG = 256 / CT;
    const long strips = (n_pixels + 7) / 8;
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_DWCONV_LN, 2.0 * n_pixels * C * KS * KS, (double)n_pixels * C * esz * 2);
    gp_timing_label("dwconv%d_ln groups C%d %dx%d B%d", KS, C, H, W, B);
    const bool narrow = dtype == GP_F16 && (cdiv(strips, PG) < dw_narrow_below() || (KS == 3 && dw3_narrow_always()));
    dim3 grid(narrow ? cdiv((n_pixels + 1) / 2, PG) : cdiv(strips, PG));
#define GP_DWG(T, K, P) hipLaunchKernelGGL((dwconv_ln_kernel<T, K, P>), grid, dim3(256), 0, s, (const T*)x, (const T*)wt, bias, ln_w, ln_b, (T*)y, H, W, C, eps, act, n_pixels, 0l, crop_group_start)
    if (dtype == GP_F16) {
        if (narrow) { if (KS == 7) GP_DWG(half_t, 7, 2); else GP_DWG(half_t, 3, 2); }
        else { if (KS == 7) GP_DWG(half_t, 7, 8); else GP_DWG(half_t, 3, 8); }
    }
    else { if (KS == 7) GP_DWG(float, 7, 8); else GP_DWG(float, 3, 8); }
#undef GP_DWG
    GP_LAUNCH_CHECK("gp_dwconv_ln_groups");
}

extern "C" int gp_dwconv7_raw_stats(const void* x, const void* wt, const float* bias, void* y, float* stats, int B, int H,
                                    int W, int C, int dtype, void* stream) {
    GP_REQUIRE(x && wt && bias && y && stats, "gp_dwconv7_raw_stats: null pointer");
    GP_REQUIRE(dtype == GP_F16, "gp_dwconv7_raw_stats: fp16 storage only");
    GP_REQUIRE(C % 128 == 0 && C >= 128 && C <= 1024, "gp_dwconv7_raw_stats: C=%d must be a multiple of 128 (<= 1024)", C);
    GP_REQUIRE(B > 0 && H % 4 == 0 && W % 16 == 0, "gp_dwconv7_raw_stats: H=%d %% 4, W=%d %% 16 required", H, W);
    hipStream_t s = (hipStream_t)stream;
    const double px = (double)B * H * W;
    gp_timing_before(s, GP_KC_DWCONV_LN, 2.0 * px * C * 49, px * C * 2 * 2);
    launch_dw7_raw(x, wt, bias, y, stats, B, H, W, C, s);
    GP_LAUNCH_CHECK("gp_dwconv7_raw_stats");
}

extern "C" int gp_layernorm(const void* x, const float* w, const float* b, void* y, long rows, int C, float eps,
                            int ldy, int dtype_in, void* stream) {
    if (ldy <= 0) ldy = C;
    GP_REQUIRE(x && w && b && y && rows > 0, "gp_layernorm: bad argument");
    const int dtype = dtype_in & ~(GP_OUT_PLANES | GP_IN_F32);
    GP_REQUIRE(dtype == GP_F32 || dtype == GP_F16, "gp_layernorm: bad dtype");
    GP_REQUIRE(!(dtype_in & GP_OUT_PLANES) || (dtype == GP_F32 && x != y && ldy == C), "gp_layernorm: GP_OUT_PLANES needs GP_F32, y != x and a dense output");
    GP_REQUIRE(!(dtype_in & GP_IN_F32) || (dtype == GP_F16 && x != y), "gp_layernorm: GP_IN_F32 (fp32 input rows) goes with GP_F16 output, y != x");
    const long pl = (dtype_in & GP_OUT_PLANES) ? rows * C : 0;
    const int esz = dtype == GP_F16 ? 2 : 4;
    GP_REQUIRE(ct_ok(C, esz), "gp_layernorm: unsupported C=%d", C);
    const int CT = C / (16 / esz), PG = 256 / CT;
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_NORM, 8.0 * rows * C, (double)rows * C * esz * 2);
    if (dtype == GP_F16 && (dtype_in & GP_IN_F32))
        hipLaunchKernelGGL((layernorm_kernel<half_t, float>), dim3(cdiv(rows, PG)), dim3(256), 0, s, (const float*)x, w, b, (half_t*)y, rows, C, eps, ldy, 0l);
    else if (dtype == GP_F16)
        hipLaunchKernelGGL(layernorm_kernel<half_t>, dim3(cdiv(rows, PG)), dim3(256), 0, s, (const half_t*)x, w, b, (half_t*)y, rows, C, eps, ldy, 0l);
    else
        hipLaunchKernelGGL(layernorm_kernel<float>, dim3(cdiv(rows, PG)), dim3(256), 0, s, (const float*)x, w, b, (float*)y, rows, C, eps, ldy, pl);
    GP_LAUNCH_CHECK("gp_layernorm");
}

extern "C" int gp_groupnorm_chunks(int B, int HW) { return cdiv(HW, gn_pxb(B, HW)); }

extern "C" int gp_groupnorm_stats(const void* x, float* partial, int B, int HW, int C, int G, int dtype,
                                  void* stream) {
    GP_REQUIRE(x && partial && B > 0 && HW > 0, "gp_groupnorm_stats: bad argument");
    GP_REQUIRE(dtype == GP_F32 || dtype == GP_F16, "gp_groupnorm_stats: bad dtype");
    const int esz = dtype == GP_F16 ? 2 : 4, vec = 16 / esz;
    GP_REQUIRE(ct_ok(C, esz) && G > 0 && G <= 256 && C % G == 0, "gp_groupnorm_stats: unsupported C=%d G=%d", C, G);
    const int cpg = C / G;
    GP_REQUIRE((cpg >= vec && cpg % vec == 0) || (cpg < vec && vec % cpg == 0 && vec / cpg <= 4),
               "gp_groupnorm_stats: channels per group %d unsupported", cpg);
    const int pxb = gn_pxb(B, HW), chunks = cdiv(HW, pxb);
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_NORM, 3.0 * B * HW * C, (double)B * HW * C * esz);
    if (dtype == GP_F16)
        hipLaunchKernelGGL(gn_partial_kernel<half_t>, dim3(chunks, B), dim3(256), 0, s, (const half_t*)x, partial, HW, C, G, pxb);
    else
        hipLaunchKernelGGL(gn_partial_kernel<float>, dim3(chunks, B), dim3(256), 0, s, (const float*)x, partial, HW, C, G, pxb);
    GP_LAUNCH_CHECK("gp_groupnorm_stats");
}

extern "C" int gp_groupnorm_apply(const void* x, const float* partial, const float* w, const float* b, void* y,
                                  int B, int HW, int C, int G, float eps, int act, int ldy, int chunks_in,
                                  int dtype_in, void* stream) {
    GP_REQUIRE(x && partial && w && b && y && B > 0 && HW > 0, "gp_groupnorm_apply: bad argument");
    const int dtype = dtype_in & ~GP_OUT_PLANES;
    GP_REQUIRE(dtype == GP_F32 || dtype == GP_F16, "gp_groupnorm_apply: bad dtype");
    GP_REQUIRE(!(dtype_in & GP_OUT_PLANES) || (dtype == GP_F32 && x != y && ldy == C), "gp_groupnorm_apply: GP_OUT_PLANES needs GP_F32, y != x and a dense output");
    const long pl = (dtype_in & GP_OUT_PLANES) ? (long)B * HW * C : 0;
    const int esz = dtype == GP_F16 ? 2 : 4;
    GP_REQUIRE(ct_ok(C, esz) && G > 0 && G <= 256 && C % G == 0, "gp_groupnorm_apply: unsupported C=%d G=%d", C, G);
    GP_REQUIRE(ldy >= C && ldy % (16 / esz) == 0, "gp_groupnorm_apply: bad ldy=%d", ldy);
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_NORM, 4.0 * B * HW * C, (double)B * HW * C * esz * 2);
    gp_timing_label("gn_apply C%d HW%d act%d", C, HW, act);
    const int chunks = chunks_in > 0 ? chunks_in : cdiv(HW, gn_pxb(B, HW)), pxb = gn_apply_pxb(B, HW);
    const float inv_count = 1.0f / ((float)HW * (C / G));
    dim3 grid(cdiv(HW, pxb), B);
    if (dtype == GP_F16)
        hipLaunchKernelGGL(gn_apply_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)x, partial, w, b, (half_t*)y, HW, C, G, act, ldy, chunks, inv_count, eps, pxb, 0l);
    else
        hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, s, (const float*)x, partial, w, b, (float*)y, HW, C, G, act, ldy, chunks, inv_count, eps, pxb, pl);
    GP_LAUNCH_CHECK("gp_groupnorm_apply");
}

/* GroupNorm apply (statistics from `partial`, as gp_groupnorm_apply) + activation + bilinear x2 upsample (align_corners) in one pass:
 * x (B, H, W, C) fp16 -> y (B, 2H, 2W, C) fp16; bitwise gp_groupnorm_apply followed by gp_upsample_bilinear2x. */
extern "C" int gp_groupnorm_upsample2x(const void* x, const float* partial, const float* w, const float* b, void* y, int B, int H,
                                       int W, int C, int G, float eps, int act, int chunks_in, int dtype, void* stream) {
    GP_REQUIRE(x && partial && w && b && y && x != y && B > 0 && H > 1 && W > 1, "gp_groupnorm_upsample2x: bad argument");
    GP_REQUIRE(dtype == GP_F16, "gp_groupnorm_upsample2x: fp16 storage only (the fp32 modes run the two passes)");
    GP_REQUIRE(ct_ok(C, 2) && C / 8 <= 64 && G > 0 && G <= 256 && C % G == 0, "gp_groupnorm_upsample2x: unsupported C=%d G=%d", C, G);
    GP_REQUIRE(B <= 65535 && 2 * H / 8 + 1 <= 65535, "gp_groupnorm_upsample2x: grid too large");
    const int HW = H * W;
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_NORM, 4.0 * B * HW * C + 8.0 * B * HW * 4 * C, (double)B * HW * C * 2 * 5);
    gp_timing_label("gn_upsample2x C%d %dx%d act%d", C, H, W, act);
    const int chunks = chunks_in > 0 ? chunks_in : cdiv(HW, gn_pxb(B, HW));
    const float inv_count = 1.0f / ((float)HW * (C / G));
    const int toy = 8;
    const int lds = (toy / 2 + 2) * 10 * C * 2;      // 30 KB at C = 256
    GP_REQUIRE(lds <= 60 * 1024, "gp_groupnorm_upsample2x: C=%d needs %d B of LDS", C, lds);
    dim3 grid(cdiv(2 * W, 16), cdiv(2 * H, toy), B);
    const int cpt = C / 8 >= 64 ? 4 : C / 8 >= 32 ? 2 : 1;
#define GP_GNUP(CPT) hipLaunchKernelGGL(gn_upsample2x_kernel<CPT>, grid, dim3(256), lds, s, (const half_t*)x, partial, w, b, (half_t*)y, H, W, C, G, act, chunks, inv_count, eps)
    if (cpt == 4) GP_GNUP(4);
    else if (cpt == 2) GP_GNUP(2);
    else GP_GNUP(1);
#undef GP_GNUP
    GP_LAUNCH_CHECK("gp_groupnorm_upsample2x");
}

extern "C" int gp_groupnorm_apply_xyz(const void* x, const float* partial, const float* w, const float* b,
                                      const float* out_w, const float* out_b, float* out_nchw, float* out_nhwc4, int B,
                                      int HW, int C, int G, float eps, int act_in, int chunks_in, int dtype, void* stream) {
    const bool h16 = (act_in & GP_ACT_PACKED16) != 0;
    const int act = act_in & 0xff;
    GP_REQUIRE(x && partial && w && b && out_w && out_b && out_nchw && out_nhwc4 && B > 0 && HW > 0, "gp_groupnorm_apply_xyz: bad argument");
    GP_REQUIRE(dtype == GP_F32 || dtype == GP_F16, "gp_groupnorm_apply_xyz: bad dtype");
    const int esz = dtype == GP_F16 ? 2 : 4;
    GP_REQUIRE(ct_ok(C, esz) && G > 0 && G <= 256 && C % G == 0, "gp_groupnorm_apply_xyz: unsupported C=%d G=%d", C, G);
    hipStream_t s = (hipStream_t)stream;
    gp_timing_before(s, GP_KC_NORM, 10.0 * B * HW * C, (double)B * HW * (C * esz + 28));
    gp_timing_label("gn_apply_xyz C%d HW%d", C, HW);
    const int chunks = chunks_in > 0 ? chunks_in : cdiv(HW, gn_pxb(B, HW)), pxb = gn_apply_pxb(B, HW);
    const float inv_count = 1.0f / ((float)HW * (C / G));
    dim3 grid(cdiv(HW, pxb), B);
    static const bool mfma = [] { const char* e = getenv("GP_GNXYZ_MFMA"); return !(e && e[0] == '0'); }();   // A/B switch
    if (dtype == GP_F16 && C == 256 && act == GP_ACT_GELU && 256 % G == 0 && mfma && h16 && gp_gelu16_enabled())
        hipLaunchKernelGGL(gn_apply_xyz_mfma_kernel<true>, grid, dim3(256), 0, s, (const half_t*)x, partial, w, b, out_w, out_b, out_nchw, out_nhwc4, HW, G, chunks, inv_count, eps, pxb);
    else if (dtype == GP_F16 && C == 256 && act == GP_ACT_GELU && 256 % G == 0 && mfma)
        hipLaunchKernelGGL(gn_apply_xyz_mfma_kernel<false>, grid, dim3(256), 0, s, (const half_t*)x, partial, w, b, out_w, out_b, out_nchw, out_nhwc4, HW, G, chunks, inv_count, eps, pxb);
    else if (dtype == GP_F16)
        hipLaunchKernelGGL(gn_apply_xyz_kernel<half_t>, grid, dim3(256), 0, s, (const half_t*)x, partial, w, b, out_w, out_b, out_nchw, out_nhwc4, HW, C, G, act, chunks, inv_count, eps, pxb);
    else
        hipLaunchKernelGGL(gn_apply_xyz_kernel<float>, grid, dim3(256), 0, s, (const float*)x, partial, w, b, out_w, out_b, out_nchw, out_nhwc4, HW, C, G, act, chunks, inv_count, eps, pxb);
    GP_LAUNCH_CHECK("gp_groupnorm_apply_xyz");
}
