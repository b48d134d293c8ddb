// Probe: what does the vector-memory path of a CU sustain for GATHERED 128-byte segments, as a function of the bytes per lane?
// The DCNv3 gather reads, per load instruction, four 128-byte segments (one (pixel, group) each) with 8 bytes per lane (16 lanes per segment).
// Arms: 8 B per lane (16 lanes per 128-byte segment, 4 segments per instruction), 16 B per lane (8 lanes per segment, 8 segments per
// instruction), 4 B per lane (32 lanes per segment, 2 segments).  Segments are picked pseudo-randomly inside a working set of WS bytes per
// workgroup-cluster: 16 KB (L1), 1 MB (L2), 512 MB (HBM / Infinity Cache).  Reports bytes per clock per CU at a nominal 2.4 GHz and
// nanoseconds per wave-level load instruction per CU.
// Build + run: hipcc --offload-arch=gfx950 -O3 scripts/probes/gather_width.hip -o gather_width && ./gather_width
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BPL>      // bytes per lane: 4, 8, 16
__global__ __launch_bounds__(256) void gather(const char* __restrict__ buf, unsigned ws_mask, int iters, float* out) {
    constexpr int LPS = 128 / BPL;               // lanes per 128-byte segment
    const int lane = threadIdx.x & 63, seg = lane / LPS, within = lane % LPS;
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned state = wid * 2654435761u + seg * 40503u + 12345u;
    float acc = 0.f;
    const char* base = buf + (size_t)(blockIdx.x % 64) * 0;     // one shared working set
    for (int i = 0; i < iters; i += 4) {
        unsigned off[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            state = state * 1664525u + 1013904223u;
            off[u] = ((state >> 4) & ws_mask & ~127u) + within * BPL;
        }
        if constexpr (BPL == 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(base + off[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
        } else if constexpr (BPL == 8) {
            float2 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float2*>(base + off[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
        } else {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float*>(base + off[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc += v[u];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

// 16 bytes per lane with SHORTER contiguous runs: SEG bytes per gathered segment (64: four lanes per segment, the row-major fragment loads of gemm_smallm_kernel; 32: two lanes)
template <int SEG>
__global__ __launch_bounds__(256) void gather_seg(const char* __restrict__ buf, unsigned ws_mask, int iters, float* out) {
    constexpr int LPS = SEG / 16;
    const int lane = threadIdx.x & 63, seg = lane / LPS, within = lane % LPS;
    const unsigned wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned state = wid * 2654435761u + seg * 40503u + 12345u;
    float acc = 0.f;
    for (int i = 0; i < iters; i += 4) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            state = state * 1664525u + 1013904223u;
            const unsigned off = ((state >> 4) & ws_mask & ~(unsigned)(SEG - 1)) + within * 16;
            v[u] = *reinterpret_cast<const float4*>(buf + off);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int SEG> int run_seg(const char* buf, size_t ws, float* out, hipStream_t s) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int G = 256 * 8, iters = 512;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(gather_seg<SEG>, dim3(G), dim3(256), 0, s, buf, (unsigned)(ws - 1), iters, out);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double instr = (double)G * 4 * iters, bytes = instr * 1024;
    printf("  16 B per lane, %3d-byte segments (%2d per instruction): %8.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)  %5.2f ns per wave-load per CU\n", SEG, 1024 / SEG, best * 1e3, bytes / best / 1e9,
           bytes / 256 / (best * 1e-3 * 2.4e9), best * 1e6 / (instr / 256));
    return 0;
}

// LDS-DMA arm: the same gathered 1 KB per instruction (8 lanes x 16 bytes per 128-byte segment), written straight into LDS (global_load_lds_dwordx4)
__global__ __launch_bounds__(256) void gather_dma(const char* __restrict__ buf, unsigned ws_mask, int iters, float* out) {
    __shared__ __attribute__((aligned(1024))) char lds[4 * 4 * 1024];          // 4 waves x 4 KB
    typedef __attribute__((address_space(3))) char lds_char_t;
    const int lane = threadIdx.x & 63, seg = lane / 8, within = lane % 8;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned wid = blockIdx.x * 4 + wave;
    unsigned state = wid * 2654435761u + seg * 40503u + 12345u;
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_char_t*)lds + wave * 4096);
    for (int i = 0; i < iters; i += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            state = state * 1664525u + 1013904223u;
            const unsigned off = ((state >> 4) & ws_mask & ~127u) + within * 16;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(off), "s"(buf), "s"(lds0 + u * 1024) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (reinterpret_cast<float*>(lds)[threadIdx.x] == 123.456f) out[0] = 1.f;
}

int run_dma(const char* buf, size_t ws, float* out, hipStream_t s) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int G = 256 * 8, iters = 512;
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(gather_dma, dim3(G), dim3(256), 0, s, buf, (unsigned)(ws - 1), iters, out);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double instr = (double)G * 4 * iters, bytes = instr * 1024;
    printf("  16 B per lane, LDS-DMA (global_load_lds_dwordx4): %8.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)  %5.2f ns per wave-load per CU\n", best * 1e3, bytes / best / 1e9,
           bytes / 256 / (best * 1e-3 * 2.4e9), best * 1e6 / (instr / 256));
    return 0;
}

template <int BPL> int run(const char* buf, size_t ws, float* out, hipStream_t s) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int G = 256 * 8, iters = 512;          // 8 workgroups of 4 waves per CU
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(gather<BPL>, dim3(G), dim3(256), 0, s, buf, (unsigned)(ws - 1), iters, out);
        CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double instr = (double)G * 4 * iters, bytes = instr * 64 * BPL;
    printf("  %2d B per lane: %8.1f us  %6.2f TB/s  %5.1f B/clk/CU (2.4 GHz)  %5.2f ns per wave-load per CU\n", BPL, best * 1e3, bytes / best / 1e9,
           bytes / 256 / (best * 1e-3 * 2.4e9), best * 1e6 / (instr / 256));
    return 0;
}

int main() {
    char* buf; float* out;
    const size_t N = 512ull << 20;
    CK(hipMalloc(&buf, N)); CK(hipMemset(buf, 0, N)); CK(hipMalloc(&out, 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (size_t ws : {16384ull, 1ull << 20, 512ull << 20}) {
        printf("working set %zu KB\n", ws >> 10);
        if (run<4>(buf, ws, out, s) || run<8>(buf, ws, out, s) || run<16>(buf, ws, out, s) || run_dma(buf, ws, out, s) || run_seg<64>(buf, ws, out, s) || run_seg<32>(buf, ws, out, s) || run_seg<256>(buf, ws, out, s)) return 1;
    }
    return 0;
}
