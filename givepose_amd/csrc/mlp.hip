// Fused ConvNeXt block MLP for gfx950 (fp16 storage):
//
//   out[m][c] = res[m][c] + gamma[c] * ( sum_h GELU( sum_k x[m][k] W1[h][k] + b1[h] ) * W2[c][h] + b2[c] )
//
// i.e. timm ConvNeXtBlock's  fc1 -> GELU -> fc2 -> gamma * . + shortcut  (network/backbone.py:36-46 builds it) in one
// launch, for C = 128 / 256 (stages 0 / 1) where the 4C-wide hidden tensor (268 / 134 MB at 64 crops) otherwise makes
// a round trip through HBM between the two GEMMs.  The hidden activations never leave the registers:
//
//   * a wave owns 32 rows (two MFMA m-tiles); its x rows are loaded once as MFMA B fragments (C/32 k-steps);
//   * the hidden dimension is walked in chunks of 32 units.  GEMM1 (A = 32 rows of W1, B = x) leaves, per lane,
//     h[m = fr][hn = nt*16 + fq*4 + j] for the two n-tiles nt of the chunk: after bias (accumulator init), GELU and
//     the conversion to fp16 those 8 values ARE the B fragment of a 16x16x32 MFMA over the chunk, with k-slot
//     fq*8 + nt*4 + j <-> hidden unit nt*16 + fq*4 + j.  The host stores W2 with its columns permuted the same way
//     (gp_convnext_mlp_pack_w2), so GEMM2's A fragment is an ordinary 16-byte read;
//   * W1 / W2 chunks (16 / 32 KB per step) stream through a 4-stage LDS ring by LDS-DMA (source-side XOR swizzle,
//     conflict-free ds_read_b128), 3 steps ahead, one barrier per chunk; weights are L2 resident;
//   * epilogue: gamma in the MFMA layout, fp16, transpose through a wave-private LDS slab, residual added with
//     packed fp16 adds, whole 256/512-byte rows stored 16 B per lane (in place over the residual is allowed).
//
// Eight waves per workgroup (256 rows), two per SIMD: the GELU of one wave (VALU) runs under the MFMAs of the other.
#include "common.hpp"

namespace {

struct MlpKP {
    const half_t* X;      // (M, C) LayerNorm output
    const half_t* W1;     // (4C, C)
    const float* b1;      // (4C)
    const half_t* W2p;    // (C, 4C), columns permuted per 32-block (see above)
    const float* b2;      // (C)
    const float* gamma;   // (C)
    const half_t* res;    // (M, C)
    half_t* out;          // (M, C)
    int M;
    int dbg;   // timing ablations (GP_MLP_DBG): 1 = no GELU, 2 = no in-loop DMA (wrong results)
};

typedef __attribute__((address_space(3))) char lds_char_t;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_addr)
                 : "memory");
}

// LDS-DMA with a wave-uniform 64-bit base (SGPR pair) and a 32-bit per-lane byte offset
__device__ __forceinline__ void glds16_sb(const void* sbase, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}

__device__ __forceinline__ f32x4 mma16(const uint4& a, const uint4& b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&a), *reinterpret_cast<const half8*>(&b), acc, 0, 0, 0);
}

// NWV = waves per workgroup (32 rows each): 8 -- one workgroup per CU -- or (C = 128, round 5; GP_MLP_WAVES=8 switches back) 4: 50 KB of LDS, so three INDEPENDENT workgroups
// share a CU instead of eight waves in lock step behind one barrier (one workgroup's GELU / epilogue beside the others' MFMAs): 205 -> 163 us per 128 crops
// The 4-wave form is trimmed to 168 registers (four GELU chains and four W2 fragments at a time) and a 3-stage ring (50 KB of LDS): THREE workgroups per CU
// (-DGP_MLP_WGS=2: the two-workgroup form with the 8-wave kernel's register choices and 4-stage ring, for A/B builds)
#ifndef GP_MLP_WGS
#define GP_MLP_WGS 3
#endif
template <int C, int G16, int NWV = 8>     // G16: GELU on packed fp16 (common.hpp gelu16_slice), the default; 0: the fp32 polynomial (GP_GELU16=0)
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? GP_MLP_WGS : 1) void convnext_mlp_kernel(const MlpKP p) {
    constexpr int HD = 4 * C, NCH = HD / 32, KS = C / 32, CT = C / 16, MT = 2;
    constexpr int ROWB = C * 2;                        // bytes per W1 row
    constexpr int W1B = 32 * ROWB, W2B = C * 64;       // bytes per chunk
    constexpr int STAGE = W1B + W2B, NS = (NWV == 4 && C == 256) ? 2 : (NWV == 4 && GP_MLP_WGS == 3) ? 3 : 4, LEAD = NS - 1;     // (C = 256 with 4 waves: a 2-stage ring = 68 KB, two workgroups per CU: measured, no gain, not instantiated)
    constexpr int I1 = W1B / 1024 / NWV, I2 = W2B / 1024 / NWV, G = I1 + I2;   // LDS-DMA instructions per wave and chunk
    constexpr int CPR1 = ROWB / 16, RPI1 = 64 / CPR1;  // 16-byte chunks per W1 row, W1 rows per DMA instruction
    constexpr int PITCH = ROWB + 16, SLAB = 32 * PITCH;
    constexpr int RING = NS * STAGE, SMEM = (RING + HD * 4) > NWV * SLAB ? (RING + HD * 4) : NWV * SLAB;
    static_assert(SMEM <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const long m0 = (long)xcd_chunk(blockIdx.x, gridDim.x) * (NWV * 32) + wave * 32;   // an XCD's workgroups = a contiguous run of rows (common.hpp)

    // ---- bias of the hidden layer -> LDS (read back per chunk as the accumulators' initial value)
    float* b1s = reinterpret_cast<float*>(smem + RING);
    for (int i = tid; i < HD / 4; i += NWV * 64) reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(p.b1)[i];

    // ---- this wave's x rows as B fragments: lane (fr, fq) holds x[m][ks*32 + fq*8 .. +8]
    uint4 xf[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[mt][ks] = *reinterpret_cast<const uint4*>(p.X + (m0 + mt * 16 + fr) * C + ks * 32 + fq * 8);

    // ---- DMA sources.  W1 chunk image: [32 rows][ROWB], 16-byte chunk ^= row & 15; W2 chunk image: [C rows][64 B],
    //      chunk ^= (-(row>>2)) & 3 inside each group of 16 rows (both involutions are repeated on the fragment reads)
    //      (32-bit per-lane byte offsets against a wave-uniform base: one register per DMA instruction instead of a 64-bit pointer)
    constexpr bool SB = true;          // (64-bit per-lane pointers: 4-8 registers more -- C = 256 with 4 waves spilled -- and 2-5 us slower at C = 128)
    unsigned w1off[I1], w2off[I2];
#pragma unroll
    for (int i = 0; i < I1; ++i) {
        const int r = (i * NWV + wave) * RPI1 + lane / CPR1, pc = lane % CPR1;
        w1off[i] = (unsigned)((r * C + ((pc ^ (r & 15)) << 3)) * 2);
    }
#pragma unroll
    for (int i = 0; i < I2; ++i) {
        const int lr = lane >> 2, r = (i * NWV + wave) * 16 + lr;
        w2off[i] = (unsigned)((r * HD + (((lane & 3) ^ ((-(lr >> 2)) & 3)) << 3)) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;
    auto stage = [&](int buf, int ch) {
        const unsigned s1 = lds0 + buf * STAGE + wave * 1024, s2 = s1 + W1B;
        const char* b1p = reinterpret_cast<const char*>(p.W1) + (long)ch * W1B;
        const char* b2p = reinterpret_cast<const char*>(p.W2p) + (long)ch * 64;
#pragma unroll
        for (int i = 0; i < I1; ++i) {
            if constexpr (SB) glds16_sb(b1p, w1off[i], s1 + i * NWV * 1024);
            else glds16(b1p + w1off[i], s1 + i * NWV * 1024);
        }
#pragma unroll
        for (int i = 0; i < I2; ++i) {
            if constexpr (SB) glds16_sb(b2p, w2off[i], s2 + i * NWV * 1024);
            else glds16(b2p + w2off[i], s2 + i * NWV * 1024);
        }
    };

    f32x4 acc2[CT][MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + ct * 16 + fq * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc2[ct][mt] = b;
    }
    __syncthreads();   // b1s visible; every compiler-visible global load above has been waited for before the DMA starts

#pragma unroll
    for (int i = 0; i < LEAD; ++i) stage(i, i);

    const int w1fo = fr * ROWB, w2fo = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
    int buf = 0, nbuf = LEAD;
    for (int ch = 0; ch < NCH; ++ch) {
        // own DMA of chunk ch landed (chunks ch+1, ch+2 stay in flight), then the barrier publishes it and retires
        // every wave's reads of chunk ch-1, whose slot the DMA of chunk ch+3 is about to overwrite
        if (LEAD >= 3 && ch + 2 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
        else if (LEAD >= 2 && ch + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (ch + LEAD < NCH && p.dbg != 2) stage(nbuf, ch + LEAD);
        const char* s1 = smem + buf * STAGE + w1fo;
        const char* s2 = smem + buf * STAGE + W1B + w2fo;

        // ---- fragment reads of the whole chunk up front (hipcc otherwise re-uses ONE fragment register set and waits
        //      lgkmcnt(0) in front of every MFMA pair); W2's land under GEMM1 and the GELU
        constexpr int GK = KS > 4 ? 1 : KS;          // k-steps of W1 fragments resident at a time (register budget: C = 256 must
                                                     // not spill -- no kernel of the path may use scratch)
        f32x4 acc1[2][MT];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(b1s + ch * 32 + nt * 16 + fq * 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc1[nt][mt] = b;
        }
        // ---- GEMM1: h[32 rows][32 hidden units of this chunk], bias as the initial value
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += GK) {
            uint4 a1[GK][2];
#pragma unroll
            for (int ks = 0; ks < GK; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    a1[ks][nt] = *reinterpret_cast<const uint4*>(s1 + nt * 16 * ROWB + ((((k0 + ks) * 4 + fq) ^ fr) << 4));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < GK; ++ks)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) acc1[nt][mt] = mma16(a1[ks][nt], xf[mt][k0 + ks], acc1[nt][mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        constexpr int GC = CT > 8 ? 2 : (NWV == 4 && GP_MLP_WGS == 3) ? 4 : CT;          // W2 fragments resident at a time
        uint4 a2[GC];
#pragma unroll
        for (int ct = 0; ct < GC; ++ct) a2[ct] = *reinterpret_cast<const uint4*>(s2 + ct * 1024);
        __builtin_amdgcn_sched_barrier(0);
        // ---- GELU on the 16 values of this lane, as 8 independent 2-element chains walked in lock step (plain v_fma_f32: the
        //      library is built without packed fp32 ops), then fp16:
        //      the B fragments of GEMM2 (k-slot fq*8 + nt*4 + j)
        f32x2 v[8];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                v[(mt * 2 + nt) * 2 + 0] = f32x2{acc1[nt][mt][0], acc1[nt][mt][1]};
                v[(mt * 2 + nt) * 2 + 1] = f32x2{acc1[nt][mt][2], acc1[nt][mt][3]};
            }
        uint4 hb[MT];
        if constexpr (G16) {
            // packed fp16: 13 operations per value PAIR, and the packed results are the dwords of GEMM2's B fragments as they stand
            unsigned hw[8];
            if (p.dbg == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) hw[i] = __builtin_bit_cast(unsigned, half2v{(half_t)v[i][0], (half_t)v[i][1]});
            } else if constexpr (C == 128 && !(NWV == 4 && GP_MLP_WGS == 3)) {
                gelu16_xn<8>(v, hw);
            } else {   // C = 256: four chains at a time (register budget)
                gelu16_xn<4>(v, hw);
                asm volatile("" : "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) : "v"(hw[0]), "v"(hw[1]), "v"(hw[2]), "v"(hw[3]));
                gelu16_xn<4>(v + 4, hw + 4);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) hb[mt] = uint4{hw[mt * 4], hw[mt * 4 + 1], hw[mt * 4 + 2], hw[mt * 4 + 3]};
        } else {
        if (p.dbg == 1) {
        } else if constexpr (C == 128) {
            gelu_poly2_x8(v);
        } else {   // C = 256 has no registers left for eight chains in flight: two at a time, each pair made to wait for
                   // the one before it (an empty asm that "writes" the next pair after "reading" the finished one)
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                gelu_poly2_xn<2>(v + i);
                if (i + 2 < 8) asm volatile("" : "+v"(v[i + 2]), "+v"(v[i + 3]) : "v"(v[i]), "v"(v[i + 1]));
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            half8 h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e * 2 + 0] = (half_t)v[mt * 4 + e][0];
                h[e * 2 + 1] = (half_t)v[mt * 4 + e][1];
            }
            hb[mt] = *reinterpret_cast<const uint4*>(&h);
        }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM2: out[32 rows][C] += h_chunk * W2p_chunk^T
#pragma unroll
        for (int c0 = 0; c0 < CT; c0 += GC) {
            if (c0 > 0) {
#pragma unroll
                for (int ct = 0; ct < GC; ++ct) a2[ct] = *reinterpret_cast<const uint4*>(s2 + (c0 + ct) * 1024);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int ct = 0; ct < GC; ++ct)
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) acc2[c0 + ct][mt] = mma16(a2[ct], hb[mt], acc2[c0 + ct][mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }
    __syncthreads();   // every wave is done with the ring: the slabs overlay it

    // ---- epilogue
    constexpr int LPR = ROWB / 16, RPS = 64 / LPR, NIT = 32 / RPS;   // lanes per row, rows per store instruction
    const int rr = lane / LPR, rc = lane % LPR;
    // C = 128: the residual rows are requested before the transpose (their latency hides under it).  C = 256: 64
    // registers of residual beside the 128 accumulators would spill, so they are requested once the accumulators are
    // in the slab (the other wave of the SIMD covers the latency); no kernel of the path may use scratch
    half8 rres[NIT];
    if constexpr (C == 128) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) rres[i] = *reinterpret_cast<const half8*>(p.res + (m0 + i * RPS + rr) * C + rc * 8);
    }
    char* slab = smem + wave * SLAB;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + ct * 16 + fq * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc2[ct][mt] * g;
            half4 o;
            for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
            *reinterpret_cast<half4*>(slab + (mt * 16 + fr) * PITCH + (ct * 4 + fq) * 8) = o;
        }
    }
    if constexpr (C != 128) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NIT; ++i) rres[i] = *reinterpret_cast<const half8*>(p.res + (m0 + i * RPS + rr) * C + rc * 8);
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        half8 v = *reinterpret_cast<const half8*>(slab + (i * RPS + rr) * PITCH + rc * 16);
        v += rres[i];
        *reinterpret_cast<half8*>(p.out + (m0 + i * RPS + rr) * C + rc * 8) = v;
    }
}

// =====================================================================================================
// The same block on v_mfma_f32_32x32x16_f16 (round 6; review item 4 of round 5): an MFMA of this shape holds the SIMD's issue port for 8 of its 32 cycles instead of 8 of 16,
// i.e. per FLOP half the MFMA issue slots, which leaves the GELU's VALU work (13 packed operations per value pair: 4 x the GELU work per FLOP of stage-2 fc1)
// 5-6 issue slots behind every MFMA instead of 1-2 (profiles/r05_mfma_valu_coissue.txt) -- at ~10 % of clock (profiles/r05_mfma_peak_bare_loop.txt).
// Layouts (lane l: j = l % 32, hh = l / 32):
//   GEMM1  D1[u][m] = sum_k W1[ch 32 + u][k] x[m][k]:  A fragment ks = W1 row j, k = 16 ks + 8 hh .. + 8 (the swizzled chunk image of the 16x16 kernel, read as
//          chunk (2 ks + hh) ^ (j & 15): conflict-free);  B fragment ks = x row j, the same k (resident);  accumulator register r = hidden unit 8 (r / 4) + 4 hh + r % 4 of row j.
//   GELU   on register pairs (r, r + 1): the packed results hw[0..7] are 16 hidden units of row j.
//   GEMM2  D2[c][m] += sum_u W2[c][u] h[m][u] in two K blocks of 16: the B fragment of block kb is hw[4 kb .. 4 kb + 3] AS IT STANDS when K slot (kb, hh, e) stands for
//          hidden unit 16 kb + 8 (e / 4) + 4 hh + e % 4: the host stores W2 with its columns in that order (gp_convnext_mlp_pack_w2_s32).  A fragment (cb, kb) = W2p row 32 cb + j,
//          16-byte chunk 2 kb + hh of the chunk's 64 bytes (the 16x16 kernel's image and swizzle).  Accumulator register r of block cb = channel 32 cb + 8 (r / 4) + 4 hh + r % 4 of row j.
// Ring, barrier, DMA, epilogue slab: as convnext_mlp_kernel.  Packed-fp16 GELU only.
typedef float mlp_f32x16 __attribute__((ext_vector_type(16)));
#ifndef GP_MLP_S32_WGS
#define GP_MLP_S32_WGS 2       // workgroups per CU of the 4-wave form (3 needs <= 168 registers: the first build spilled 8)
#endif
template <int C, int NWV>
__global__ __launch_bounds__(NWV * 64, NWV == 4 ? GP_MLP_S32_WGS : 1) void convnext_mlp_s32_kernel(const MlpKP p) {
    constexpr int HD = 4 * C, NCH = HD / 32, KS = C / 16, CB = C / 32;
    constexpr int ROWB = C * 2;
    constexpr int W1B = 32 * ROWB, W2B = C * 64;
    constexpr int STAGE = W1B + W2B, NS = (NWV == 4 && GP_MLP_S32_WGS == 3) ? 3 : 4, LEAD = NS - 1;
    constexpr int I1 = W1B / 1024 / NWV, I2 = W2B / 1024 / NWV, G = I1 + I2;
    constexpr int CPR1 = ROWB / 16, RPI1 = 64 / CPR1;
    constexpr int PITCH = ROWB + 16, SLAB = 32 * PITCH;
    constexpr int RING = NS * STAGE, SMEM = (RING + HD * 4) > NWV * SLAB ? (RING + HD * 4) : NWV * SLAB;
    static_assert(SMEM <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const long m0 = (long)xcd_chunk(blockIdx.x, gridDim.x) * (NWV * 32) + wave * 32;

    float* b1s = reinterpret_cast<float*>(smem + RING);
    for (int i = tid; i < HD / 4; i += NWV * 64) reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(p.b1)[i];

    uint4 xf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) xf[ks] = *reinterpret_cast<const uint4*>(p.X + (m0 + j) * C + ks * 16 + hh * 8);

    unsigned w1off[I1], w2off[I2];
#pragma unroll
    for (int i = 0; i < I1; ++i) {
        const int r = (i * NWV + wave) * RPI1 + lane / CPR1, pc = lane % CPR1;
        w1off[i] = (unsigned)((r * C + ((pc ^ (r & 15)) << 3)) * 2);
    }
#pragma unroll
    for (int i = 0; i < I2; ++i) {
        const int lr = lane >> 2, r = (i * NWV + wave) * 16 + lr;
        w2off[i] = (unsigned)((r * HD + (((lane & 3) ^ ((-(lr >> 2)) & 3)) << 3)) * 2);
    }
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;
    auto stage = [&](int buf, int ch) {
        const unsigned s1 = lds0 + buf * STAGE + wave * 1024, s2 = s1 + W1B;
        const char* b1p = reinterpret_cast<const char*>(p.W1) + (long)ch * W1B;
        const char* b2p = reinterpret_cast<const char*>(p.W2p) + (long)ch * 64;
#pragma unroll
        for (int i = 0; i < I1; ++i) glds16_sb(b1p, w1off[i], s1 + i * NWV * 1024);
#pragma unroll
        for (int i = 0; i < I2; ++i) glds16_sb(b2p, w2off[i], s2 + i * NWV * 1024);
    };

    mlp_f32x16 acc2[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + cb * 32 + g * 8 + hh * 4);
            acc2[cb][4 * g] = b[0]; acc2[cb][4 * g + 1] = b[1]; acc2[cb][4 * g + 2] = b[2]; acc2[cb][4 * g + 3] = b[3];
        }
    __syncthreads();

#pragma unroll
    for (int i = 0; i < LEAD; ++i) stage(i, i);

    const int w1fo = j * ROWB, w2fo = j * 64, w2sw = (-((j & 15) >> 2)) & 3;
    int buf = 0, nbuf = LEAD;
    for (int ch = 0; ch < NCH; ++ch) {
        if (LEAD >= 3 && ch + 2 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
        else if (LEAD >= 2 && ch + 1 < NCH) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (ch + LEAD < NCH && p.dbg != 2) stage(nbuf, ch + LEAD);
        const char* s1 = smem + buf * STAGE + w1fo;
        const char* s2 = smem + buf * STAGE + W1B + w2fo;

        mlp_f32x16 acc1;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(b1s + ch * 32 + g * 8 + hh * 4);
            acc1[4 * g] = b[0]; acc1[4 * g + 1] = b[1]; acc1[4 * g + 2] = b[2]; acc1[4 * g + 3] = b[3];
        }
        // ---- GEMM1, W1 fragments four k-steps at a time
        constexpr int GK = C == 128 ? 2 : 1;       // (register budget: 168 at three workgroups per CU / 256 with 128 accumulators + 64 of x)
#pragma unroll
        for (int k0 = 0; k0 < KS; k0 += GK) {
            uint4 a1[GK];
#pragma unroll
            for (int ks = 0; ks < GK; ++ks) a1[ks] = *reinterpret_cast<const uint4*>(s1 + (((2 * (k0 + ks) + hh) ^ (j & 15)) << 4));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < GK; ++ks)
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(&a1[ks]), *reinterpret_cast<const half8*>(&xf[k0 + ks]), acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- W2 fragments of K block 0 land under the GELU
        constexpr int GC = 2;
        uint4 a2[GC];
#pragma unroll
        for (int cb = 0; cb < GC; ++cb) a2[cb] = *reinterpret_cast<const uint4*>(s2 + cb * 2048 + ((hh ^ w2sw) << 4));
        __builtin_amdgcn_sched_barrier(0);
        unsigned hw[8];
        if (p.dbg == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) hw[i] = __builtin_bit_cast(unsigned, half2v{(half_t)acc1[2 * i], (half_t)acc1[2 * i + 1]});
        } else {        // four chains at a time (register budget)
            f32x2 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = f32x2{acc1[2 * i], acc1[2 * i + 1]};
            gelu16_xn<4>(v, hw);
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = f32x2{acc1[8 + 2 * i], acc1[8 + 2 * i + 1]};
            asm volatile("" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) : "v"(hw[0]), "v"(hw[1]), "v"(hw[2]), "v"(hw[3]));
            gelu16_xn<4>(v, hw + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- GEMM2
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            const uint4 hb = uint4{hw[4 * kb], hw[4 * kb + 1], hw[4 * kb + 2], hw[4 * kb + 3]};
#pragma unroll
            for (int c0 = 0; c0 < CB; c0 += GC) {
                if (kb > 0 || c0 > 0) {
#pragma unroll
                    for (int cb = 0; cb < GC; ++cb) a2[cb] = *reinterpret_cast<const uint4*>(s2 + (c0 + cb) * 2048 + (((2 * kb + hh) ^ w2sw) << 4));
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int cb = 0; cb < GC; ++cb)
                    acc2[c0 + cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8*>(&a2[cb]), *reinterpret_cast<const half8*>(&hb), acc2[c0 + cb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        buf = buf + 1 == NS ? 0 : buf + 1;
        nbuf = nbuf + 1 == NS ? 0 : nbuf + 1;
    }
    __syncthreads();

    // ---- epilogue: gamma, fp16, transpose through the wave's slab, residual, whole rows
    constexpr int LPR = ROWB / 16, RPS = 64 / LPR, NIT = 32 / RPS;
    const int rr = lane / LPR, rc = lane % LPR;
    half8 rres[NIT];
    if constexpr (C == 128) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) rres[i] = *reinterpret_cast<const half8*>(p.res + (m0 + i * RPS + rr) * C + rc * 8);
    }
    char* slab = smem + wave * SLAB;
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c = cb * 32 + g * 8 + hh * 4;
            const f32x4 gm = *reinterpret_cast<const f32x4*>(p.gamma + c);
            half4 o;
            o[0] = (half_t)(acc2[cb][4 * g] * gm[0]); o[1] = (half_t)(acc2[cb][4 * g + 1] * gm[1]);
            o[2] = (half_t)(acc2[cb][4 * g + 2] * gm[2]); o[3] = (half_t)(acc2[cb][4 * g + 3] * gm[3]);
            *reinterpret_cast<half4*>(slab + j * PITCH + c * 2) = o;
        }
    if constexpr (C != 128) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NIT; ++i) rres[i] = *reinterpret_cast<const half8*>(p.res + (m0 + i * RPS + rr) * C + rc * 8);
    }
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        half8 v = *reinterpret_cast<const half8*>(slab + (i * RPS + rr) * PITCH + rc * 16);
        v += rres[i];
        *reinterpret_cast<half8*>(p.out + (m0 + i * RPS + rr) * C + rc * 8) = v;
    }
}

// =====================================================================================================
// The same block for C = 512 (ConvNeXt stage 2: 27 of the 36 blocks; round 5), ONE wave per SIMD.
// The two-launch path runs fc1 on the weights-in-registers kernel and fc2 on the ping-pong tile kernel: 134 MB of hidden activations
// (128 crops) are stored and fetched again, both kernels pay a prologue / epilogue with the matrix pipe idle, and fc2's 256 epilogues hit
// HBM in lock step.  Here a workgroup of FOUR waves (256 threads: up to 512 registers per wave) owns 128 rows: a wave keeps its 32 x rows as
// B fragments (128 registers) and the whole 32 x 512 output tile as accumulators (256 registers), and walks the 2048 hidden units in 64
// chunks of 32: GEMM1 (64 MFMAs: A = W1 fragments from LDS) -> GELU (packed fp16, common.hpp) -> GEMM2 (64 MFMAs: A = W2p fragments, B = the
// GELU output as it stands in registers).  With one wave per SIMD nothing else hides a latency, so the loop is software-pipelined INSIDE the
// wave: iteration c runs GEMM1 of chunk c+1 with the GELU of chunk c riding behind its MFMAs (two hidden accumulator sets), then GEMM2 of
// chunk c; fragment reads are issued one MFMA group ahead.
// Weights stream through a 4-slot LDS ring of 32 KB PIECES in consumption order -- W1[0], then W1[c+1], W2[c] for every c -- by LDS-DMA, three
// pieces ahead, one barrier per piece in the MIDDLE of the phase before it (mid() below: RAW for the next piece, WAR for the slot of the previous one).
__global__ __launch_bounds__(256) void convnext_mlp512_kernel(const MlpKP p) {
    constexpr int C = 512, HD = 2048, NCH = HD / 32, KS = C / 32, CT = C / 16, MT = 2, NW = 4;
    constexpr int ROWB = C * 2, PIECE = 32768, NP = 4, LEAD = 3, NPIECES = 2 * NCH, IPW = PIECE / 1024 / NW;   // 8 DMA instructions per wave and piece
    constexpr int PITCH = ROWB + 16, SLAB = 32 * PITCH;
    constexpr int RING = NP * PIECE, SMEM = (RING + HD * 4) > NW * SLAB ? (RING + HD * 4) : NW * SLAB;
    static_assert(SMEM <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(1024))) char smem[SMEM];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
    const long m0 = (long)xcd_chunk(blockIdx.x, gridDim.x) * 128 + wave * 32;

    float* b1s = reinterpret_cast<float*>(smem + RING);
    for (int i = tid; i < HD / 4; i += 256) reinterpret_cast<f32x4*>(b1s)[i] = reinterpret_cast<const f32x4*>(p.b1)[i];

    // ---- this wave's x rows as B fragments: lane (fr, fq) holds x[m][ks*32 + fq*8 .. +8]
    uint4 xf[MT][KS];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            xf[mt][ks] = *reinterpret_cast<const uint4*>(p.X + (m0 + mt * 16 + fr) * C + ks * 32 + fq * 8);

    // ---- DMA sources.  W1 piece image: [32 rows][1 KB], 16-byte chunk ^= row & 15 (one instruction = one row); W2 piece image:
    //      [512 rows][64 B], chunk ^= (-(row >> 2)) & 3 inside each group of 16 rows (one instruction = 16 rows)
    //      Instruction i of a wave: W1 row 4 i + wave, chunk lane ^ ((4 i + wave) & 15) = (lane ^ wave) ^ (4 (i & 3)) (wave < 4): ONE offset register,
    //      the row term 4 i KB goes into the scalar base; W2 rows (4 i + wave) * 16 + lane / 4: likewise one register + a scalar term.
    const unsigned w1off = (unsigned)(wave * ROWB + ((lane ^ wave) << 4));
    const int lr = lane >> 2;
    const unsigned w2off = (unsigned)(((wave * 16 + lr) * HD + (((lane & 3) ^ ((-(lr >> 2)) & 3)) << 3)) * 2);
    const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem;
    // piece q of the stream: q = 0 -> W1[0]; odd q < NPIECES - 1 -> W1[(q + 1) / 2]; even q >= 2 -> W2[q / 2 - 1]; q = NPIECES - 1 -> W2[NCH - 1]
    auto stage_instr = [&](int q, int i) {      // DMA instruction i (of IPW) of this wave for piece q
        const unsigned d = lds0 + (q & (NP - 1)) * PIECE + wave * 1024 + i * NW * 1024;
        const bool is_w1 = q == 0 || ((q & 1) && q != NPIECES - 1);
        if (is_w1) {
            const int ch = (q + 1) >> 1;
            glds16_sb(reinterpret_cast<const char*>(p.W1) + (long)ch * 32 * ROWB + i * NW * ROWB, w1off ^ (unsigned)((4 * (i & 3)) << 4), d);
        } else {
            const int ch = q == NPIECES - 1 ? NCH - 1 : (q >> 1) - 1;
            glds16_sb(reinterpret_cast<const char*>(p.W2p) + (long)ch * 64 + (long)i * NW * 16 * HD * 2, w2off, d);
        }
    };
    // Synchronisation, ONE barrier per piece, in the MIDDLE of the phase that consumes piece q (mid(q)): the wave waits for ITS DMA of piece q+1
    // (counted vmcnt: piece q+2, always the one piece issued after it, stays in flight), the barrier publishes piece q+1 -- so phase q+1 starts
    // without a barrier, its first fragment reads right behind the last MFMAs of phase q -- and, every wave being at least half way through
    // phase q, ends all reads of piece q-1, whose slot piece q+3 may then take.
    auto mid = [&](int q) {
        if (q + 1 >= NPIECES) return;
        __builtin_amdgcn_sched_barrier(0);
        if (q + 2 < NPIECES) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    // The DMA of piece q+3 rides in the second half of phase q: its source (W1 or W2p chunk) is picked with scalar selects, not branches -- control
    // flow inside the unrolled MFMA stream made hipcc spill 84 registers -- and the phases that have nothing left to fetch are separate
    // instantiations (RF = false).
    struct Refill { const char* base; long stride; unsigned voff, xm, dst; };
    auto refill_of = [&](int qq) {
        const bool is_w1 = qq == 0 || ((qq & 1) && qq != NPIECES - 1);
        const int ch = is_w1 ? (qq + 1) >> 1 : (qq == NPIECES - 1 ? NCH - 1 : (qq >> 1) - 1);
        Refill r;
        r.base = is_w1 ? reinterpret_cast<const char*>(p.W1) + (long)ch * 32 * ROWB : reinterpret_cast<const char*>(p.W2p) + (long)ch * 64;
        r.stride = is_w1 ? (long)NW * ROWB : (long)NW * 16 * HD * 2;
        r.voff = is_w1 ? w1off : w2off;
        r.xm = is_w1 ? ~0u : 0u;
        r.dst = lds0 + (qq & (NP - 1)) * PIECE + wave * 1024;
        return r;
    };
    auto refill = [&](const Refill& r, int i) {
#ifndef GP_MLP512_NODMA     // investigation build: the loop without its in-loop LDS-DMA (wrong results): what the DMA issue costs a lone wave
        glds16_sb(r.base + i * r.stride, r.voff ^ ((unsigned)((4 * (i & 3)) << 4) & r.xm), r.dst + i * NW * 1024);
#endif
    };

    f32x4 acc2[CT][MT];
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(p.b2 + ct * 16 + fq * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc2[ct][mt] = b;
    }
    __syncthreads();   // b1s visible; every compiler-visible global load above has been waited for before the DMA starts
#pragma unroll
    for (int q = 0; q < LEAD; ++q)
#pragma unroll
        for (int i = 0; i < IPW; ++i) stage_instr(q, i);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * IPW) : "memory");   // piece 0 (pieces 1, 2 stay in flight)
    __builtin_amdgcn_s_barrier();

    const int w1fo = fr * ROWB, w2fo = fr * 64 + ((fq ^ ((-(fr >> 2)) & 3)) << 4);
    f32x4 acc1[2][MT];           // [nt][mt]: the hidden accumulators of the chunk GEMM1 is working on
    unsigned xq[8];              // the PREVIOUS chunk's 16 hidden values as 8 packed fp16 pairs (what the GELU in GEMM1's shadow reads):
                                 // pair c = mt * 4 + nt * 2 + e <-> registers 2 e, 2 e + 1 of acc1[nt][mt] = dword c % 4 of GEMM2's B fragment hb[mt]
    unsigned hw[8];              // ... and their GELU, packed: the dwords of hb
    uint4 hb[MT];
    half2v hx[2], hu[2], hp[2];

    // GEMM1 of chunk ch (W1 piece in ring slot `slot`); SH: the GELU of the previous chunk (xq -> hw) rides behind its MFMAs.  At the end the
    // chunk's own values are rounded to packed fp16 into xq (the GELU's first step: 8 conversions outside the shadow buy 8 registers --
    // one hidden accumulator set instead of two; the kernel sits at the 512-register limit).
    auto gemm1 = [&](int ch, int q, auto shc, auto rf1c, auto rf2c) {
        constexpr bool RF1 = decltype(rf1c)::value, RF2 = decltype(rf2c)::value;      // fetch piece q+2 in the first half / piece q+3 in the second
        const int slot = q & (NP - 1);
        const Refill rp1 = refill_of(RF1 ? q + 2 : 0), rp2 = refill_of(RF2 ? q + 3 : 0);
        constexpr bool SH = decltype(shc)::value;
        constexpr int NSL = GELU16_SLICES - 1, NCALL = 8 * NSL, NMF = KS * 4;      // slices 1 .. 12 (slice 0, the conversion, is done)
        const char* s1 = smem + slot * PIECE + w1fo;
        uint4 a1[3][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) a1[ks][nt] = *reinterpret_cast<const uint4*>(s1 + nt * 16 * ROWB + (((ks * 4 + fq) ^ fr) << 4));
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(b1s + ch * 32 + nt * 16 + fq * 4);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc1[nt][mt] = b;
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, KS>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks + 2 < KS) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) a1[(ks + 2) % 3][nt] = *reinterpret_cast<const uint4*>(s1 + nt * 16 * ROWB + ((((ks + 2) * 4 + fq) ^ fr) << 4));
            }
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 4>([&](auto jc) {
                constexpr int j = decltype(jc)::value, nt = j / 2, mt = j % 2, i = ks * 4 + j;
                if constexpr (i == NMF / 2) mid(q);
                acc1[nt][mt] = mma16(a1[ks % 3][nt], xf[mt][ks], acc1[nt][mt]);
                if constexpr (RF1 && i < NMF / 2 && i % (NMF / 2 / IPW) == 0) refill(rp1, i / (NMF / 2 / IPW));
                if constexpr (RF2 && i >= NMF / 2 && (i - NMF / 2) % (NMF / 2 / IPW) == 0) refill(rp2, (i - NMF / 2) / (NMF / 2 / IPW));
                if constexpr (SH) {
                    constexpr int n0 = i * NCALL / NMF, n1 = (i + 1) * NCALL / NMF;
                    static_for<n0, n1>([&](auto nc) {
                        constexpr int n = decltype(nc)::value, g = n / (2 * NSL), m = n % (2 * NSL), slot_ = m / 2 + 1, chn = m % 2, c = 2 * g + chn;
                        if constexpr (slot_ == 1) hx[chn] = __builtin_bit_cast(half2v, xq[c]);
                        gelu16_slice<slot_>(f32x2{0.f, 0.f}, hx[chn], hu[chn], hp[chn]);
                        if constexpr (slot_ == GELU16_SLICES - 1) hw[c] = __builtin_bit_cast(unsigned, hp[chn]);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        static_for<0, 8>([&](auto cc) {
            constexpr int c = decltype(cc)::value, pmt = c / 4, pnt = (c / 2) & 1, pe = c & 1;
            xq[c] = __builtin_bit_cast(unsigned, half2v{(half_t)acc1[pnt][pmt][2 * pe], (half_t)acc1[pnt][pmt][2 * pe + 1]});
        });
    };
    // GELU of xq with nothing to hide behind (only the last chunk needs it)
    auto gelu_plain = [&]() {
        static_for<0, 8>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            half2v xh = __builtin_bit_cast(half2v, xq[c]), u, pp;
            static_for<1, GELU16_SLICES>([&](auto sc) { gelu16_slice<decltype(sc)::value>(f32x2{0.f, 0.f}, xh, u, pp); });
            hw[c] = __builtin_bit_cast(unsigned, pp);
        });
    };
    // GEMM2 of the chunk whose GELU output sits in hw (W2 piece in ring slot `slot`)
    auto gemm2 = [&](int q) {
        const int slot = q & (NP - 1);
        const char* s2 = smem + slot * PIECE + w2fo;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) hb[mt] = uint4{hw[mt * 4], hw[mt * 4 + 1], hw[mt * 4 + 2], hw[mt * 4 + 3]};
        uint4 a2[4];
#pragma unroll
        for (int ct = 0; ct < 3; ++ct) a2[ct] = *reinterpret_cast<const uint4*>(s2 + ct * 1024);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, CT>([&](auto ctc) {
            constexpr int ct = decltype(ctc)::value;
            if constexpr (ct + 3 < CT) a2[(ct + 3) & 3] = *reinterpret_cast<const uint4*>(s2 + (ct + 3) * 1024);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (ct == CT / 2) mid(q);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) acc2[ct][mt] = mma16(a2[ct & 3], hb[mt], acc2[ct][mt]);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    using T_ = std::true_type;
    using F_ = std::false_type;
    // piece 0: W1[0] (pieces 0, 1, 2 are in flight from the prologue)
    gemm1(0, 0, F_{}, F_{}, F_{});
    // chunks c = 0 .. NCH-2: piece 2c+1 = W1[c+1] (GEMM1 of chunk c+1 with the GELU of chunk c behind it), piece 2c+2 = W2[c] (GEMM2 of chunk c).
    // ALL DMA rides in the GEMM1 phases (an LDS-DMA asm inside GEMM2's MFMA stream made hipcc spill 77 registers): phase q = 2c+1 fetches piece
    // q+2 in its first half (slot of piece q-2: every wave is past phase q-1) and piece q+3 behind its barrier (slot of piece q-1).
    for (int c = 0; c + 2 < NCH; ++c) {
        gemm1(c + 1, 2 * c + 1, T_{}, T_{}, T_{});
        gemm2(2 * c + 2);
    }
    gemm1(NCH - 1, NPIECES - 3, T_{}, T_{}, F_{});       // the last piece (W2[NCH-1]) in its first half, nothing after that
    gemm2(NPIECES - 2);
    // the last chunk: its GELU has no GEMM1 to ride behind; its W2 piece is the last of the stream
    gelu_plain();
    gemm2(NPIECES - 1);
    __syncthreads();   // every wave is done with the ring: the slabs overlay it

    // ---- epilogue: gamma in the MFMA layout, fp16, transpose through a wave-private LDS slab, + residual, 1 KB rows stored 16 B per lane
    char* slab = smem + wave * SLAB;
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(p.gamma + ct * 16 + fq * 4);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 v = acc2[ct][mt] * g;
            half4 o;
            for (int j = 0; j < 4; ++j) o[j] = (half_t)v[j];
            *reinterpret_cast<half4*>(slab + (mt * 16 + fr) * PITCH + (ct * 4 + fq) * 8) = o;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i0 = 0; i0 < 32; i0 += 8) {
        half8 rres[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) rres[i] = *reinterpret_cast<const half8*>(p.res + (m0 + i0 + i) * C + lane * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            half8 v = *reinterpret_cast<const half8*>(slab + (i0 + i) * PITCH + lane * 16);
            v += rres[i];
            *reinterpret_cast<half8*>(p.out + (m0 + i0 + i) * C + lane * 8) = v;
        }
    }
}

// W2 (C, 4C) -> W2p: inside every block of 32 hidden units, k-slot s = fq*8 + nt*4 + j takes unit nt*16 + fq*4 + j
__global__ void mlp_pack_w2_kernel(const half_t* w2, half_t* w2p, int C, int HD) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)C * HD) return;
    const int c = (int)(i / HD), s = (int)(i - (long)c * HD);
    const int blk = s >> 5, t = s & 31, fq = t >> 3, nt = (t >> 2) & 1, j = t & 3;
    w2p[i] = w2[(long)c * HD + blk * 32 + nt * 16 + fq * 4 + j];
}

// the 32x32x16 kernel's order: inside every block of 32 hidden units, K slot s = kb*16 + hh*8 + e takes unit 16 kb + 8 (e / 4) + 4 hh + e % 4
__global__ void mlp_pack_w2_s32_kernel(const half_t* w2, half_t* w2p, int C, int HD) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)C * HD) return;
    const int c = (int)(i / HD), s = (int)(i - (long)c * HD);
    const int blk = s >> 5, t = s & 31, kb = t >> 4, hh = (t >> 3) & 1, e = t & 7;
    w2p[i] = w2[(long)c * HD + blk * 32 + 16 * kb + 8 * (e >> 2) + 4 * hh + (e & 3)];
}

}  // namespace

extern "C" int gp_convnext_mlp_pack_w2_s32(const void* w2, void* w2p, int C, void* stream) {
    GP_REQUIRE(w2 && w2p && w2 != w2p, "gp_convnext_mlp_pack_w2_s32: bad pointers");
    GP_REQUIRE(C == 128 || C == 256, "gp_convnext_mlp_pack_w2_s32: C=%d must be 128 or 256", C);
    const long n = (long)C * 4 * C;
    hipLaunchKernelGGL(mlp_pack_w2_s32_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(w2), reinterpret_cast<half_t*>(w2p), C, 4 * C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gp_fail(GP_ERR_LAUNCH, "gp_convnext_mlp_pack_w2_s32: %s", hipGetErrorString(e));
    return GP_OK;
}

extern "C" int gp_convnext_mlp_pack_w2(const void* w2, void* w2p, int C, void* stream) {
    GP_REQUIRE(w2 && w2p && w2 != w2p, "gp_convnext_mlp_pack_w2: bad pointers");
    GP_REQUIRE(C == 128 || C == 256 || C == 512, "gp_convnext_mlp_pack_w2: C=%d must be 128, 256 or 512", C);
    const long n = (long)C * 4 * C;
    hipLaunchKernelGGL(mlp_pack_w2_kernel, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const half_t*>(w2), reinterpret_cast<half_t*>(w2p), C, 4 * C);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gp_fail(GP_ERR_LAUNCH, "gp_convnext_mlp_pack_w2: %s", hipGetErrorString(e));
    return GP_OK;
}

extern "C" int gp_convnext_mlp(const void* x, const void* w1, const float* b1, const void* w2p, const float* b2,
                               const float* gamma, const void* residual, void* out, long M, int C, int dtype_in, void* stream) {
    const bool s32 = (dtype_in & GP_MLP_S32) != 0;     // w2p in gp_convnext_mlp_pack_w2_s32's order: the 32x32x16 kernel
    const int dtype = dtype_in & ~GP_MLP_S32;
    GP_REQUIRE(dtype == GP_F16, "gp_convnext_mlp: fp16 storage only (fp32 runs fc1 / fc2 through gp_gemm)");
    GP_REQUIRE(!s32 || (C == 128 && gp_gelu16_enabled()), "gp_convnext_mlp: GP_MLP_S32 exists for C = 128 with the packed-fp16 GELU");
    GP_REQUIRE(C == 128 || C == 256 || C == 512, "gp_convnext_mlp: C=%d must be 128, 256 or 512", C);
    GP_REQUIRE(x && w1 && b1 && w2p && b2 && gamma && residual && out, "gp_convnext_mlp: null operand");
    GP_REQUIRE(M > 0 && M % (C == 512 ? 128 : 256) == 0, "gp_convnext_mlp: M=%ld must be a positive multiple of %d", M, C == 512 ? 128 : 256);
    GP_REQUIRE((((size_t)x | (size_t)w1 | (size_t)w2p | (size_t)residual | (size_t)out | (size_t)b1 | (size_t)b2 | (size_t)gamma) & 15) == 0,
               "gp_convnext_mlp: operands must be 16-byte aligned");
    GP_REQUIRE(x != out, "gp_convnext_mlp: x and out must not alias (out may alias residual)");
    MlpKP p;
    p.X = reinterpret_cast<const half_t*>(x); p.W1 = reinterpret_cast<const half_t*>(w1); p.b1 = b1;
    p.W2p = reinterpret_cast<const half_t*>(w2p); p.b2 = b2; p.gamma = gamma;
    p.res = reinterpret_cast<const half_t*>(residual); p.out = reinterpret_cast<half_t*>(out); p.M = (int)M;
    { const char* e = getenv("GP_MLP_DBG"); p.dbg = e ? atoi(e) : 0; }
    hipStream_t s = (hipStream_t)stream;
    const double flops = 2.0 * 2.0 * (double)M * C * 4 * C;
    const double bytes = 3.0 * M * C * 2 + 2.0 * 4 * C * C * 2;
    gp_timing_before(s, GP_KC_GEMM, flops, bytes);
    gp_timing_label("convnext_mlp C%d M%ld", C, M);
    if (C == 512) {
        const dim3 grid512((unsigned)(M / 128));
        GP_REQUIRE(gp_gelu16_enabled(), "gp_convnext_mlp: the C = 512 kernel exists with the packed-fp16 GELU only (GP_GELU16=0: run fc1 / fc2 through gp_gemm)");
        hipLaunchKernelGGL(convnext_mlp512_kernel, grid512, dim3(256), 0, s, p);
        GP_LAUNCH_CHECK("gp_convnext_mlp");
    }
    if (s32) {
        gp_timing_label("convnext_mlp s32 C%d M%ld", C, M);
        GP_REQUIRE(C == 128, "gp_convnext_mlp: GP_MLP_S32 is built for C = 128 (C = 256 needs 128 accumulator + 64 x registers beside a 16-register GEMM1 accumulator: it spilled 55)");
        hipLaunchKernelGGL((convnext_mlp_s32_kernel<128, 4>), dim3((unsigned)(M / 128)), dim3(256), 0, s, p);
        GP_LAUNCH_CHECK("gp_convnext_mlp");
    }
    const dim3 grid((unsigned)(M / 256));
    // C = 128 (round 5): 4-wave workgroups, three per CU (50 KB of LDS each): 205 -> 163 us per 128 crops against eight waves behind one barrier
    // (profiles/r05_mlp_waves_ab.txt; C = 256's ring leaves no room for a second workgroup -- with a 2-stage ring it has, and gains nothing).  GP_MLP_WAVES=8: A/B switch
    static const int mlp_waves = [] { const char* e = getenv("GP_MLP_WAVES"); return e ? atoi(e) : 4; }();
    if (gp_gelu16_enabled() && C == 128 && mlp_waves == 4) {
        hipLaunchKernelGGL((convnext_mlp_kernel<128, 1, 4>), dim3((unsigned)(M / 128)), dim3(256), 0, s, p);
        GP_LAUNCH_CHECK("gp_convnext_mlp");
    }
    if (gp_gelu16_enabled()) {
        if (C == 128) hipLaunchKernelGGL((convnext_mlp_kernel<128, 1>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((convnext_mlp_kernel<256, 1>), grid, dim3(512), 0, s, p);
    } else {
        if (C == 128) hipLaunchKernelGGL((convnext_mlp_kernel<128, 0>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((convnext_mlp_kernel<256, 0>), grid, dim3(512), 0, s, p);
    }
    GP_LAUNCH_CHECK("gp_convnext_mlp");
}
