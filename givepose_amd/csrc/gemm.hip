// MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[m][n] = epi( sum_k X[m][k] * W[n][k] + bias[n] )
//
// Tile 128 (m) x 128 (n) x 128 bytes of K per step, 4 waves (2 m x 2 n), each wave 64x64 as 4x4
// MFMA 16x16 tiles.  W rows are the MFMA "A" operand and X rows the "B" operand, so a lane's four
// accumulator registers are four consecutive n of one m: the epilogue stores 8 B (f16) / 16 B (f32)
// contiguous per lane straight from registers.  Both operands are staged through LDS as
// [row][128 B] with a 16-byte-chunk XOR swizzle (chunk ^= row & 7) that makes the ds_read_b128
// fragment reads bank-conflict free (lane groups of 16 hit 16 distinct 16-B slots).  Global->LDS is
// register staged and double buffered: tile t+1 is in flight while tile t is multiplied, one
// barrier per K step.  In conv mode the X loader walks (kh,kw,ci) in K order and zero-fills the
// padding, so no im2col buffer ever exists in HBM.  Workgroup ids are remapped so that each XCD
// (ids equal mod 8 share one) owns a contiguous run of tiles with n fastest: the X panel of an
// m-tile is fetched into one XCD's L2 once and reused by all its n-tiles.
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
};

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
__global__ __launch_bounds__(256, 2) void gemm_kernel(const GemmKP p) {
    constexpr int EPT = 16 / sizeof(T);   // elements per 16-byte chunk
    constexpr int KPT = 128 / sizeof(T);  // K elements per step
    __shared__ __attribute__((aligned(16))) char smem[2 * 32768];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;

    // XCD-chunked, bijective tile remap (ids equal mod 8 share an XCD)
    const int nblk = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int q = nblk >> 3, r8 = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    const int tile = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + idx;
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int m0 = tm * 128, n0 = tn * 128;
    const int kt_begin = blockIdx.y * p.kt_per_split;
    const int kt_end = min(p.nkt, kt_begin + p.kt_per_split);

    const T* __restrict__ X = reinterpret_cast<const T*>(p.X);
    const T* __restrict__ W = reinterpret_cast<const T*>(p.W);

    // ---- loader state: thread owns 16-B chunk `lc` of rows lr, lr+32, lr+64, lr+96 of both tiles
    const int lc = tid & 7, lr = tid >> 3;
    long xbase[4];
    int hi0[4], wi0[4];
    bool xok[4], wok[4];
    const T* wrow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + lr + 32 * i;
        xok[i] = m < p.M;
        if (p.conv) {
            const int hw = p.Ho * p.Wo;
            const int b = m / hw, r = m - b * hw;
            const int ho = r / p.Wo, wo = r - ho * p.Wo;
            hi0[i] = ho * p.stride - p.pad;
            wi0[i] = wo * p.stride - p.pad;
            xbase[i] = (long)b * p.H * p.Win;
        } else {
            hi0[i] = wi0[i] = 0;
            xbase[i] = (long)m * p.ldx + lc * EPT;
        }
        const int n = n0 + lr + 32 * i;
        wok[i] = n < p.N;
        wrow[i] = W + (long)n * p.K + lc * EPT;
    }
    const int cpt = p.conv ? p.Cin / KPT : 1;  // k-steps per filter tap

    uint4 xr[4], wr[4];
    auto gload = [&](int kt) {
        int kh = 0, kw = 0, ci = 0;
        if (p.conv) {
            const int tap = kt / cpt;
            ci = (kt - tap * cpt) * KPT + lc * EPT;
            kh = tap / p.KW;
            kw = tap - kh * p.KW;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (p.conv) {
                const int hi = hi0[i] + kh, wi = wi0[i] + kw;
                if (xok[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.Win)
                    v = *reinterpret_cast<const uint4*>(X + (xbase[i] + (long)hi * p.Win + wi) * p.Cin + ci);
            } else if (xok[i]) {
                v = *reinterpret_cast<const uint4*>(X + xbase[i] + (long)kt * KPT);
            }
            xr[i] = v;
            wr[i] = wok[i] ? *reinterpret_cast<const uint4*>(wrow[i] + (long)kt * KPT) : make_uint4(0, 0, 0, 0);
        }
    };
    const int soff = lr * 128 + ((lc ^ (lr & 7)) << 4);
    auto sstore = [&](int buf) {
        char* xs = smem + buf * 32768;
        char* ws = xs + 16384;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<uint4*>(xs + soff + i * 4096) = xr[i];
            *reinterpret_cast<uint4*>(ws + soff + i * 4096) = wr[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    const int xfo = (wm * 64 + fr) * 128, wfo = (wn * 64 + fr) * 128, sw = fr & 7;
    auto compute = [&](int buf) {
        const char* xs = smem + buf * 32768;
        const char* ws = xs + 16384;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int co = ((ks * 4 + fq) ^ sw) << 4;
            uint4 xf[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                xf[t] = *reinterpret_cast<const uint4*>(xs + xfo + t * 2048 + co);
                wf[t] = *reinterpret_cast<const uint4*>(ws + wfo + t * 2048 + co);
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) mma<T>(acc[nt][mt], wf[nt], xf[mt]);
        }
    };

    if (kt_begin < kt_end) {
        gload(kt_begin);
        sstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) gload(kt + 1);
        compute(buf);
        if (more) sstore(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue straight from registers: lane holds C[m][n..n+3]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + fr;
        if (m >= p.M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + fq * 4;
            if (n >= p.N) continue;
            if (p.splitk > 1)
                *reinterpret_cast<f32x4*>(p.ws + ((long)blockIdx.y * p.M + m) * p.N + n) = acc[nt][mt];
            else
                epi_store<T>(p, m, n, acc[nt][mt]);
        }
    }
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

}  // namespace

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
    GP_REQUIRE(d->epilogue >= GP_EPI_NONE && d->epilogue <= GP_EPI_SCALE_RES, "gp_gemm: bad epilogue");
    if (d->epilogue == GP_EPI_SCALE_RES)
        GP_REQUIRE(d->gamma && d->residual && d->ldres % 4 == 0 && d->ldres >= d->N, "gp_gemm: SCALE_RES needs gamma/residual");
    GemmKP p;
    memset(&p, 0, sizeof(p));
    p.X = d->X; p.W = d->W; p.bias = d->bias; p.gamma = d->gamma; p.res = d->residual; p.C = d->C; p.ws = d->workspace;
    p.M = d->M; p.N = d->N; p.K = d->K; p.ldx = d->ldx; p.ldc = d->ldc; p.ldres = d->ldres;
    p.epi = d->epilogue; p.out_f32 = d->out_f32;
    if (d->KH > 0) {
        GP_REQUIRE(d->KW > 0 && d->stride > 0 && d->pad >= 0, "gp_gemm: bad conv geometry");
        GP_REQUIRE(d->Cin % KPT == 0, "gp_gemm: conv Cin=%d must be a multiple of %d", d->Cin, KPT);
        GP_REQUIRE(d->K == d->KH * d->KW * d->Cin, "gp_gemm: conv K=%d != KH*KW*Cin", d->K);
        const int Ho = (d->H + 2 * d->pad - d->KH) / d->stride + 1, Wo = (d->Win + 2 * d->pad - d->KW) / d->stride + 1;
        GP_REQUIRE(Ho == d->Ho && Wo == d->Wo, "gp_gemm: conv output %dx%d != expected %dx%d", d->Ho, d->Wo, Ho, Wo);
        GP_REQUIRE((long)d->B * Ho * Wo == d->M, "gp_gemm: conv M=%d != B*Ho*Wo", d->M);
        p.conv = 1; p.H = d->H; p.Win = d->Win; p.Cin = d->Cin; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
        p.Ho = Ho; p.Wo = Wo;
    } else {
        GP_REQUIRE(d->ldx >= d->K && d->ldx % (16 / esz) == 0, "gp_gemm: ldx=%d invalid", d->ldx);
    }
    p.tiles_m = cdiv(d->M, 128);
    p.tiles_n = cdiv(d->N, 128);
    p.nkt = d->K / KPT;
    p.splitk = d->splitk > 1 ? d->splitk : 1;
    if (p.splitk > p.nkt) p.splitk = p.nkt;
    p.kt_per_split = cdiv(p.nkt, p.splitk);
    p.splitk = cdiv(p.nkt, p.kt_per_split);
    if (p.splitk > 1) GP_REQUIRE(d->workspace != nullptr, "gp_gemm: splitk needs a workspace");
    hipStream_t s = (hipStream_t)stream;
    const double flops = 2.0 * d->M * d->N * d->K;
    const double xbytes = d->KH > 0 ? (double)d->B * d->H * d->Win * d->Cin * esz : (double)d->M * d->K * esz;
    const double bytes = xbytes + (double)d->N * d->K * esz + (double)d->M * d->N * (d->out_f32 ? 4 : esz) +
                         (d->epilogue == GP_EPI_SCALE_RES ? (double)d->M * d->N * esz : 0.0);
    gp_timing_before(s, GP_KC_GEMM, flops, bytes);
    dim3 grid(p.tiles_m * p.tiles_n, p.splitk);
    if (d->dtype == GP_F16)
        hipLaunchKernelGGL(gemm_kernel<half_t>, grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL(gemm_kernel<float>, grid, dim3(256), 0, s, p);
    if (p.splitk > 1) {
        const long work = (long)d->M * (d->N / 4);
        if (d->dtype == GP_F16)
            hipLaunchKernelGGL(splitk_reduce_kernel<half_t>, dim3(cdiv(work, 256)), dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL(splitk_reduce_kernel<float>, dim3(cdiv(work, 256)), dim3(256), 0, s, p);
    }
    GP_LAUNCH_CHECK("gp_gemm");
}
