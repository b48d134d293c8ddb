// How fast can a CU pull GEMM operand rows through LDS-DMA (global_load_lds_dwordx4) as a function of how the 64 x 16 bytes of one wave
// instruction are laid over the rows?  The tile kernels of csrc/gemm.hip fetch SEG = 64 bytes per row and K step (ping-pong GEMM, 3x3
// window conv) or 128 (the generic ring kernels); the small-M kernel measured 16-byte pieces at 15 B/clk/CU and 64-byte pieces at
// 19-31.  Here: one workgroup per CU (8 waves, as the 256 x 256 tiles), each walks its own panel of ROWS rows
// x 4096 bytes (K = 2048 halfs: stage-2 fc2) K step by K step into a 64 KB LDS ring, nothing consumes the data; SEG = 64 / 128 /
// 256 / 1024 bytes per row per instruction.  The panel set (256 x 512 rows x 4 KB = 512 MB... too big for L2, like the real operands:
// X from HBM / MALL, W (2 MB) from L2) is approximated by: X panel private per workgroup (streamed), W panel shared by all.
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 scripts/dma_segment_bench.hip -o /tmp/dma_seg && /tmp/dma_seg
#include <hip/hip_runtime.h>
#include <stdio.h>

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_addr) : "memory");
}

template <int SEG>
__global__ __launch_bounds__(512) void fetch_kernel(const char* x, const char* w, int rows_x, int rows_w, int pitch, int reps, unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    constexpr int LPR = SEG / 16, RPI = 64 / LPR;            // lanes per row, rows per instruction
    const int rl = lane / LPR, cl = lane % LPR;
    const char* xp = x + (long)blockIdx.x * rows_x * pitch;   // this workgroup's X panel
    typedef __attribute__((address_space(3))) char lds_char_t;
    const unsigned base = (unsigned)(size_t)(lds_char_t*)lds + wave * 8192;            // 8 KB of the ring per wave: 8 instructions in flight
    const int steps = pitch / SEG;
    unsigned slot = 0;
    for (int rep = 0; rep < reps; ++rep)
        for (int k = 0; k < steps; ++k) {
            // X rows [wave * rows_x / 8, ...) and W rows likewise: each wave covers its share of the rows, RPI rows per instruction
            for (int r0 = wave * (rows_x / 8); r0 < (wave + 1) * (rows_x / 8); r0 += RPI) {
                glds16(xp + (long)(r0 + rl) * pitch + k * SEG + cl * 16, base + (slot & 7) * 1024);
                ++slot;
                if ((slot & 7) == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
            for (int r0 = wave * (rows_w / 8); r0 < (wave + 1) * (rows_w / 8); r0 += RPI) {
                glds16(w + (long)(r0 + rl) * pitch + k * SEG + cl * 16, base + (slot & 7) * 1024);
                ++slot;
                if ((slot & 7) == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && lds[0] == 0x7f && out) out[0] = 1;
}

template <int SEG> static void run(const char* x, const char* w, unsigned* out, int wgs) {
    const int rows_x = 256, rows_w = 256, pitch = 4096, reps = 4;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(fetch_kernel<SEG>, dim3(wgs), dim3(512), 65536, 0, x, w, rows_x, rows_w, pitch, reps, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (it && ms < best) best = ms;
    }
    const double bytes_per_wg = (double)(rows_x + rows_w) * pitch * reps;
    printf("SEG %4d B per row and instruction (%2d rows per instruction): %7.1f us for %.1f MB per workgroup -> %6.1f GB/s per CU = %5.1f B/clk at 2.4 GHz; %5.2f TB/s over %d CUs\n",
           SEG, 64 / (SEG / 16), best * 1e3, bytes_per_wg / 1e6, bytes_per_wg / (best * 1e-3) / 1e9, bytes_per_wg / (best * 1e-3) / 2.4e9,
           bytes_per_wg * wgs / (best * 1e-3) / 1e12, wgs);
}

int main() {
    const int wgs = 256;
    char *x, *w; unsigned* out;
    hipMalloc(&x, (size_t)wgs * 256 * 4096); hipMalloc(&w, (size_t)256 * 4096); hipMalloc(&out, 4);
    hipMemset(x, 1, (size_t)wgs * 256 * 4096); hipMemset(w, 1, (size_t)256 * 4096); hipMemset(out, 0, 4);
    
    hipDeviceSynchronize();
    printf("one workgroup (8 waves) per CU; X panel 256 rows x 4 KB private per workgroup (256 MB in all: MALL / HBM), W panel 256 rows x 4 KB shared (L2); 4 passes\n");
    run<64>(x, w, out, wgs);
    run<128>(x, w, out, wgs);
    run<256>(x, w, out, wgs);
    run<1024>(x, w, out, wgs);
    run<64>(x, w, out, wgs);
    return 0;
}
