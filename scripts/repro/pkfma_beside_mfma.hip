// Stand-alone reproducer (plain HIP, no inline asm, no product code) of the round-1 "batches in flight" corruption on
// MI355X (gfx950, ROCm 7.2): DESIGN.md section 6b.
//
//   stream A: `aggressor`  -- the round-1 register-staged 128x128x(128 B) MFMA GEMM tile loop (global_load_dwordx4 ->
//                             ds_write_b128, then 8 ds_read_b128 + 16 v_mfma_f32_16x16x32_f16 per half K step, as hipcc
//                             schedules them), M = 64, N = 2048, K = 8192 (PoseNet's PnP fc1 shape): 16 workgroups.
//   stream B: `victim`     -- out[row][c] = w0[c]*p[row].x + w1[c]*p[row].y + w2[c]*p[row].z + b[c] into fp16 (the
//                             MAPEncoder's first 1x1 conv on the xyz map); hipcc emits v_pk_fma_f32 with op_sel.
// The two kernels share no memory.  The victim's output is compared bitwise (on the device) with its own result
// computed before the aggressor was ever launched.  Observed: 3-48 % of the victim launches come back with 16
// elements wrong in one row -- one 16-lane pass (lanes 16-31 or 48-63) of ONE packed-fp32 result register, low half.
// With the victim's FMAs forced to scalar v_fma_f32 (-DNOPK) or the aggressor's MFMA loop removed (-DNOMFMA): 0.
//
//   hipcc -O3 --offload-arch=gfx950 -o pkfma_beside_mfma scripts/repro/pkfma_beside_mfma.hip && ./pkfma_beside_mfma [rounds]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifndef MTN
#define MTN 4
#endif
#ifndef VPXT
#define VPXT 8
#endif
#ifndef ACC0
#define ACC0 0.f      /* -DACC0=1e30f: if the victim's wrong values become huge, accumulator data of the aggressor leaks into it */
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

// ------------------------------------------------------------------------------------------------ aggressor
#ifdef LB1
#define AGG_LB __launch_bounds__(256)
#else
#define AGG_LB __launch_bounds__(256, 2)
#endif
__global__ AGG_LB void aggressor(const half_t* __restrict__ X, const half_t* __restrict__ W, half_t* __restrict__ C,
                                                    int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) char smem[2 * 32768];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave & 1, wn = wave >> 1;
    const int tiles_n = (N + 127) / 128;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm * 128, n0 = tn * 128, nkt = K / 64;
    const int lc = tid & 7, lr = tid >> 3;
    uint4 xr[4], wr[4];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + lr + 32 * i, n = n0 + lr + 32 * i;
            xr[i] = m < M ? *reinterpret_cast<const uint4*>(X + (long)m * K + kt * 64 + lc * 8) : make_uint4(0, 0, 0, 0);
            wr[i] = n < N ? *reinterpret_cast<const uint4*>(W + (long)n * K + kt * 64 + lc * 8) : make_uint4(0, 0, 0, 0);
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
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{ACC0, ACC0, ACC0, ACC0};
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
                for (int mt = 0; mt < MTN; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(*reinterpret_cast<const half8*>(&wf[nt]), *reinterpret_cast<const half8*>(&xf[mt]),
                                                                         acc[nt][mt], 0, 0, 0);
        }
    };
    gload(0);
    sstore(0);
    __syncthreads();
    int buf = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        const bool more = kt + 1 < nkt;
#ifndef NOLOADS
        if (more) gload(kt + 1);
#endif
#ifndef NOMFMA
        compute(buf);
#endif
#ifndef NOLOADS
        if (more) sstore(buf ^ 1);
#endif
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const int m = m0 + wm * 64 + mt * 16 + fr;
        if (m >= M) continue;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int n = n0 + wn * 64 + nt * 16 + fq * 4;
            if (n >= N) continue;
            half4 o;
            for (int j = 0; j < 4; ++j) o[j] = (half_t)(acc[nt][mt][j] + (float)xr[0].x * 0.f);
            *reinterpret_cast<half4*>(C + (long)m * N + n) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ victim
__global__ __launch_bounds__(256) void victim(const float* __restrict__ xyz4, const float* __restrict__ w, const float* __restrict__ bias,
                                              half_t* __restrict__ y, long rows, int Cout) {
    constexpr int VEC = 8, PXT = VPXT;
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
        half8 o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
#if defined(PKH)
            // round 5: is the PACKED FP16 family (v_pk_fma_f16, which the fp16 GELU of the GEMM kernels now uses) disturbed like v_pk_fma_f32?
            // Two channels per instruction, operands rounded to fp16 first; compared bitwise with the same arithmetic run solo.
            if (e & 1) continue;
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            h2 t = {(half_t)bv[e], (half_t)bv[e + 1]};
            t = __builtin_elementwise_fma(h2{(half_t)w2[e], (half_t)w2[e + 1]}, h2{(half_t)p[2], (half_t)p[2]}, t);
            t = __builtin_elementwise_fma(h2{(half_t)w1[e], (half_t)w1[e + 1]}, h2{(half_t)p[1], (half_t)p[1]}, t);
            t = __builtin_elementwise_fma(h2{(half_t)w0[e], (half_t)w0[e + 1]}, h2{(half_t)p[0], (half_t)p[0]}, t);
            o[e] = t[0]; o[e + 1] = t[1];
#elif defined(NOPK)
            float t = bv[e];
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(w2[e]), "v"(p[2]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(w1[e]), "v"(p[1]));
            asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(t) : "v"(w0[e]), "v"(p[0]));
            o[e] = (half_t)t;
#else
            o[e] = (half_t)fmaf(w0[e], p[0], fmaf(w1[e], p[1], fmaf(w2[e], p[2], bv[e])));
#endif
        }
        *reinterpret_cast<half8*>(y + row * Cout + cs * VEC) = o;
    }
}

__global__ void fill_f32(float* p, long n, unsigned seed) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { unsigned s = (unsigned)i * 2654435761u + seed; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; p[i] = (float)(s & 0xFFFF) / 32768.f - 1.f; }
}
__global__ void fill_f16(half_t* p, long n, unsigned seed, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) { unsigned s = (unsigned)i * 2654435761u + seed; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13; p[i] = (half_t)(((float)(s & 0xFFFF) / 32768.f - 1.f) * scale); }
}
__global__ void count_diff(const unsigned short* a, const unsigned short* b, long n, unsigned* cnt) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n && a[i] != b[i]) atomicAdd(cnt, 1u);
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 100, NV = 16, NA = 12;
    const long R = 262144; const int CO = 256, M = 64, N = 2048, K = 8192;
    float *xyz4, *w, *b; half_t *ref, *out[16], *ax, *aw, *ac; unsigned* cnt;
    CK(hipMalloc(&xyz4, R * 4 * 4)); CK(hipMalloc(&w, CO * 3 * 4)); CK(hipMalloc(&b, CO * 4)); CK(hipMalloc(&ref, R * CO * 2));
    for (int i = 0; i < NV; ++i) CK(hipMalloc(&out[i], R * CO * 2));
    CK(hipMalloc(&ax, (long)M * K * 2)); CK(hipMalloc(&aw, (long)N * K * 2)); CK(hipMalloc(&ac, (long)M * N * 2)); CK(hipMalloc(&cnt, 4 * NV));
    fill_f32<<<(R * 4 + 255) / 256, 256>>>(xyz4, R * 4, 1); fill_f32<<<3, 256>>>(w, CO * 3, 2); fill_f32<<<1, 256>>>(b, CO, 3);
    fill_f16<<<((long)M * K + 255) / 256, 256>>>(ax, (long)M * K, 4, 1.f); fill_f16<<<((long)N * K + 255) / 256, 256>>>(aw, (long)N * K, 5, 0.01f);
    const int vgrid = (int)(R / (8 * VPXT));
    victim<<<vgrid, 256>>>(xyz4, w, b, ref, R, CO);
    CK(hipDeviceSynchronize());
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));
    long bad = 0, total = 0;
    for (int r = 0; r < rounds; ++r) {
        CK(hipMemsetAsync(cnt, 0, 4 * NV, sb));
        for (int i = 0; i < NA; ++i) aggressor<<<16, 256, 0, sa>>>(ax, aw, ac, M, N, K);
        for (int i = 0; i < NV; ++i) victim<<<vgrid, 256, 0, sb>>>(xyz4, w, b, out[i], R, CO);
        CK(hipDeviceSynchronize());
        for (int i = 0; i < NV; ++i) count_diff<<<(R * CO + 255) / 256, 256, 0, sb>>>((const unsigned short*)out[i], (const unsigned short*)ref, R * CO, cnt + i);
        unsigned h[16];
        CK(hipMemcpyAsync(h, cnt, 4 * NV, hipMemcpyDeviceToHost, sb));
        CK(hipStreamSynchronize(sb));
        for (int i = 0; i < NV; ++i) {
            total++;
            if (!h[i]) continue;
            bad++;
            if (bad <= 3) {
                printf("round %d victim launch %d: %u fp16 elements differ from the victim's own solo result\n", r, i, h[i]);
                static half_t *ho = nullptr, *hr = nullptr;
                if (!ho) { ho = (half_t*)malloc(R * CO * 2); hr = (half_t*)malloc(R * CO * 2); CK(hipMemcpy(hr, ref, R * CO * 2, hipMemcpyDeviceToHost)); }
                CK(hipMemcpy(ho, out[i], R * CO * 2, hipMemcpyDeviceToHost));
                int shown = 0;
                for (long e = 0; e < R * CO && shown < 16; ++e)
                    if (((unsigned short*)ho)[e] != ((unsigned short*)hr)[e]) { printf("   row %ld col %ld: got %g, solo %g\n", e / CO, e % CO, (double)ho[e], (double)hr[e]); ++shown; }
            }
        }
    }
    printf("%ld corrupted victim launches of %ld\n", bad, total);
    return bad ? 1 : 0;
}
