// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access shapes this library uses (MI355X_MICROARCH.md, HBM section:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B / lane) ... other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern").  Three kernels read a KNOWN number of bytes from a
// 1 GiB buffer (far beyond the 256 MiB Infinity Cache), each byte once:
//   rd16   : streaming, 16 B per lane (the LDS-DMA / GEMM shape)
//   rd8    : streaming, 8 B per lane  (half4 loads: norm kernels in fp16)
//   rows8  : 128-byte rows at pseudo-random row indices, 16 lanes x 8 B per row (the DCNv3 gather's corner fetch)
// Run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- ./pmc_calib`; scripts/pmc_calib.py divides.
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void rd16(const uint4* p, long n, unsigned* out) {
    uint4 a = {0, 0, 0, 0};
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const uint4 v = p[i]; a.x ^= v.x; a.y ^= v.y; a.z ^= v.z; a.w ^= v.w; }
    if ((a.x ^ a.y ^ a.z ^ a.w) == 0x12345678u) out[0] = 1;
}
__global__ void rd8(const uint2* p, long n, unsigned* out) {
    uint2 a = {0, 0};
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256) { const uint2 v = p[i]; a.x ^= v.x; a.y ^= v.y; }
    if ((a.x ^ a.y) == 0x12345678u) out[0] = 1;
}
__global__ void rows8(const uint2* p, long nrows, long nfetch, unsigned* out) {   // row = bijective scramble of the fetch index
    uint2 a = {0, 0};
    for (long g = (blockIdx.x * 256L + threadIdx.x) >> 4; g < nfetch; g += (long)gridDim.x * 16) {
        const long row = (g * 2654435761L + 12345L) % nrows;     // nrows is a power of two times an odd factor: multiplicative scramble
        const uint2 v = p[row * 16 + (threadIdx.x & 15)];
        a.x ^= v.x; a.y ^= v.y;
    }
    if ((a.x ^ a.y) == 0x12345678u) out[0] = 1;
}

int main() {
    const long bytes = 1L << 30;
    void* buf; unsigned* out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4);
    hipMemset(buf, 1, bytes); hipMemset(out, 0, 4);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(rd16, dim3(8192), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, out);
        hipLaunchKernelGGL(rd8, dim3(8192), dim3(256), 0, 0, (const uint2*)buf, bytes / 8, out);
        hipLaunchKernelGGL(rows8, dim3(8192), dim3(256), 0, 0, (const uint2*)buf, bytes / 128, bytes / 128, out);   // every row once, scrambled order
        hipDeviceSynchronize();
    }
    printf("known bytes per launch: rd16 %ld rd8 %ld rows8 %ld\n", bytes, bytes, bytes);
    return 0;
}
