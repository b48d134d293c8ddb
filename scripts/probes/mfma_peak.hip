// Round-5 probe: what a BARE fp16 MFMA loop sustains on MI355X -- operands in registers, no LDS, no memory -- on random and on zero operands: TFLOP/s over the whole chip
// and the shader clock the chip holds meanwhile (s_memtime / s_memrealtime).  The "roofline" an MFMA kernel can be priced against on real data.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_peak scripts/probes/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 0: 16x16x32, 1: 32x32x16
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* stamps, int iters, int zero) {
    half8 a[8], b[8];
    unsigned s = (blockIdx.x * 512 + threadIdx.x) * 2654435761u + 12345u;
    for (int r = 0; r < 8; ++r)
        for (int i = 0; i < 8; ++i) {
            s = s * 1664525u + 1013904223u; const float fa = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
            s = s * 1664525u + 1013904223u; const float fb = ((s >> 8) & 0xFFFF) / 32768.0f - 1.0f;
            a[r][i] = zero ? (_Float16)0.f : (_Float16)fa;
            b[r][i] = zero ? (_Float16)0.f : (_Float16)(fb * 0.05f);
        }
    f32x4 c4[8] = {};
    f32x16 c16[2] = {};
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it += 8) {      // (the rotation of the B operands is unrolled: a run-time register index would go through scratch)
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (SHAPE == 0) c4[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[u], b[(u + r) & 7], c4[u], 0, 0, 0);
                else c16[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u], b[(u + r) & 7], c16[u & 1], 0, 0, 0);
            }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0;
    for (int u = 0; u < 8; ++u) acc += c4[u][0] + c4[u][3];
    for (int i = 0; i < 16; ++i) acc += c16[0][i] + c16[1][i];
    out[blockIdx.x * 512 + threadIdx.x] = acc;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
void run(int zero, int threads, float* out, unsigned long long* st) {
    const int blocks = 256, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // ~2 s of back-to-back launches first (the clock settles), then time 10 launches
    for (int i = 0; i < 150; ++i) k<SHAPE><<<blocks, threads>>>(out, st, iters, zero);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) k<SHAPE><<<blocks, threads>>>(out, st, iters, zero);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[512];
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double clk = 0; for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 0.1; clk /= blocks;
    const double flop = (SHAPE ? 32768.0 : 16384.0) * 8.0 * iters * (threads / 64) * blocks * 10;
    fflush(stdout);
    printf("%s  %s operands  %d wave(s) per SIMD: %7.1f TFLOP/s  in-kernel clock %.3f GHz  (%.0f FLOP per cycle and CU; the pipes' ceiling: 4096)\n", SHAPE ? "32x32x16" : "16x16x32",
           zero ? "zero  " : "random", threads / 256, flop / (ms * 1e-3) / 1e12, clk, flop / 10 / blocks / ((double)h[0]));
}

int main() {
    float* out; unsigned long long* st;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&st, 512 * 8);
    for (int zero = 0; zero < 2; ++zero)
        for (int threads : {256, 512}) { run<0>(zero, threads, out, st); run<1>(zero, threads, out, st); }
    return 0;
}
