// Round-5 probe: which instructions of the PARTNER wave take issue slots from a wave's MFMA stream on the same SIMD (gfx950).
// 512-thread workgroups (two waves per SIMD): waves 0-3 run MFMAs only (16x16x32, eight accumulators); waves 4-7 run a stream of ONE instruction kind
// (nothing / v_fma_f32 / ds_read_b128 / global_load_lds_dwordx4 (LDS-DMA, 1 KB, L2-resident source) / global_load_dwordx4 / s_nop-only scalar work) with a counted wait so that
// ~N of them issue per MFMA of the partner.  Reported: cycles per MFMA of waves 0-3 (16 = the pipe) and the partner's instructions per 1000 cycles.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/steal scripts/probes/mfma_partner_steal.hip && /tmp/steal
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char_t;

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, const char* src, unsigned long long* st, int iters, unsigned* flag_dummy) {
    __shared__ __attribute__((aligned(1024))) char smem[65536];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = uint4{1u, 2u, 3u, 4u};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wave < 4) {
        half8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(0.02f * i - lane * 0.001f); }
        f32x4 c[8] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c[u], 0, 0, 0);
        }
        float s = 0;
        for (int u = 0; u < 8; ++u) s += c[u][0] + c[u][2];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) st[blockIdx.x * 16 + wave] = t1 - t0;
    } else {
        // the partner: run until the MFMA waves are (about) done: a fixed count sized by the host per kind
        const unsigned lds0 = (unsigned)(size_t)(lds_char_t*)smem + (wave - 4) * 8192;
        const char* g = src + (size_t)blockIdx.x * 65536 + (wave - 4) * 8192 + lane * 16;
        float f = lane, fm = 1.0001f, fa = 0.5f;
        asm volatile("" : "+v"(fm), "+v"(fa));
        uint4 acc = {0, 0, 0, 0};
        unsigned long long n = 0;
        const int reps = iters * (KIND == 0 ? 0 : 1);
        for (int it = 0; it < reps; ++it) {
            if constexpr (KIND == 1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f) : "v"(fm), "v"(fa));
            } else if constexpr (KIND == 2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { uint4 v = *reinterpret_cast<const uint4*>(smem + (wave - 4) * 8192 + q * 1024 + lane * 16); asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w)); }
            } else if constexpr (KIND == 3) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g + q * 1024), "s"(lds0 + q * 1024) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else if constexpr (KIND == 4) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { uint4 v = *reinterpret_cast<const uint4*>(g + q * 1024); acc.x ^= v.x; }
            } else if constexpr (KIND == 5) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
            }
            n += 8;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) { st[blockIdx.x * 16 + wave] = t1 - t0; st[blockIdx.x * 16 + 8 + wave - 4] = n; }
        if (f == 12345.f || acc.x == 0x12345u) flag_dummy[0] = 1;
    }
}

template <int KIND>
void run(const char* name, float* out, const char* src, unsigned long long* st, unsigned* fd) {
    const int blocks = 256, iters = 4000;
    hipMemset(st, 0, blocks * 16 * 8);
    k<KIND><<<blocks, 512>>>(out, src, st, iters, fd);
    k<KIND><<<blocks, 512>>>(out, src, st, iters, fd);
    hipDeviceSynchronize();
    static unsigned long long h[256 * 16];
    hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost);
    double m = 0, pt = 0, pn = 0;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) { m += h[b * 16 + w]; pt += h[b * 16 + 4 + w]; pn += h[b * 16 + 8 + w]; }
    m /= blocks * 4; pt /= blocks * 4; pn /= blocks * 4;
    printf("partner: %-28s MFMA waves %6.2f cycles per MFMA (pipe: 16)   partner issued %7.0f instructions in %9.0f cycles = one per %6.1f cycles; while the MFMA waves ran: %5.2f per MFMA\n",
           name, m / (iters * 8.0), pn, pt, pn > 0 ? pt / pn : 0.0, pn > 0 ? (pn / pt) * (m / (iters * 8.0)) : 0.0);
}

int main() {
    float* out; char* src; unsigned long long* st; unsigned* fd;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&src, 256 * 65536 + 65536); hipMalloc(&st, 256 * 16 * 8); hipMalloc(&fd, 4);
    hipMemset(src, 1, 256 * 65536 + 65536);
    run<0>("nothing (idle partner)", out, src, st, fd);
    run<1>("v_fma_f32", out, src, st, fd);
    run<2>("ds_read_b128", out, src, st, fd);
    run<3>("global_load_lds_dwordx4 (DMA)", out, src, st, fd);
    run<4>("global_load_dwordx4", out, src, st, fd);
    run<5>("4 x s_nop", out, src, st, fd);
    return 0;
}
