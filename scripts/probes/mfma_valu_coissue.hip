// Round-5 probe: how many VALU instructions hide behind an MFMA on gfx950, with ONE and with TWO waves per SIMD.
// Each wave runs N iterations of [1 MFMA + F independent VALU fillers]; cycles per MFMA from s_memtime (wave 0 of each block).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/coissue scripts/probes/mfma_valu_coissue.hip && /tmp/coissue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE, int F, int KIND>   // SHAPE 0: 16x16x32, 1: 32x32x16.  KIND 0: v_fma_f32, 1: v_pk_fma_f16
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(threadIdx.x * 0.002f - i); }
    f32x4 c4[4] = {};
    f32x16 c16 = {};
    float f[8];
    half2v hf[8];
    for (int i = 0; i < 8; ++i) { f[i] = threadIdx.x + i; hf[i] = half2v{(_Float16)(i + 1), (_Float16)(i + 2)}; }
    const float m = 1.0001f, ad = 0.5f;
    const half2v hm = {(_Float16)1.001f, (_Float16)1.001f}, ha = {(_Float16)0.5f, (_Float16)0.5f};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (SHAPE == 0) c4[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c4[u & 3], 0, 0, 0);
            else c16 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c16, 0, 0, 0);
#pragma unroll
            for (int q = 0; q < F; ++q) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[q & 7]) : "v"(m), "v"(ad));
                else asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(hf[q & 7]) : "v"(hm), "v"(ha));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 4; ++i) s += c4[i][0] + c4[i][3];
    for (int i = 0; i < 16; ++i) s += c16[i];
    for (int i = 0; i < 8; ++i) s += f[i] + (float)hf[i][0] + (float)hf[i][1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int SHAPE, int F, int KIND>
void run(int threads, float* out, unsigned long long* cyc) {
    const int iters = 2000, blocks = 256;
    hipMemset(cyc, 0, blocks * 8 * 8);
    k<SHAPE, F, KIND><<<blocks, threads>>>(out, cyc, iters);
    k<SHAPE, F, KIND><<<blocks, threads>>>(out, cyc, iters);
    hipDeviceSynchronize();
    static unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double mx = 0, sum = 0; int n = 0;
    for (int i = 0; i < blocks * 8; ++i) if (h[i]) { sum += h[i]; n++; if (h[i] > mx) mx = h[i]; }
    printf("%s  %d wave(s)/SIMD  %d x %s per MFMA: %6.1f cycles per MFMA and wave (mean; slowest wave %6.1f) -> per SIMD %.1f cycles per MFMA issued\n", SHAPE ? "32x32x16" : "16x16x32", threads / 256,
           F, KIND ? "v_pk_fma_f16" : "v_fma_f32", sum / n / (iters * 8.0), mx / (iters * 8.0), sum / n / (iters * 8.0) / (threads / 256));
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
#define ROW(S, F, KD) run<S, F, KD>(256, out, cyc); run<S, F, KD>(512, out, cyc);
    ROW(0, 0, 0) ROW(0, 1, 0) ROW(0, 2, 0) ROW(0, 3, 0) ROW(0, 4, 0) ROW(0, 6, 0)
    ROW(0, 1, 1) ROW(0, 2, 1) ROW(0, 3, 1) ROW(0, 4, 1)
    ROW(1, 0, 0) ROW(1, 2, 0) ROW(1, 4, 0) ROW(1, 6, 0) ROW(1, 8, 0) ROW(1, 12, 0)
    ROW(1, 2, 1) ROW(1, 4, 1) ROW(1, 6, 1) ROW(1, 8, 1)
    return 0;
}
