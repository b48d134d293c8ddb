// Probe: what does a grid barrier by atomic counter cost on MI355X (agent-scope release / acquire, bounded spin), against the ~4.6 us of a
// dependent kernel launch inside a hipGraph?  G workgroups x 256 threads, NB barriers in sequence; every workgroup writes a line before each
// barrier and reads another workgroup's line behind it (checks visibility across XCDs).
// Build + run: hipcc --offload-arch=gfx950 -O3 scripts/probes/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* fail) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 2000000) { ok = false; *fail = 1; break; }      // bounded: every wave reaches the exit
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void probe(unsigned* counter, unsigned* data, int nb, int* fail, unsigned* bad) {
    const int G = gridDim.x, g = blockIdx.x;
    for (int b = 0; b < nb; ++b) {
        data[(size_t)(b & 1) * G * 256 + g * 256 + threadIdx.x] = (unsigned)(b * 1000003 + g * 256 + threadIdx.x);
        if (!grid_barrier(counter, (unsigned)(b + 1) * G, fail)) return;
        const int o = (g + G / 2 + b) % G;      // some workgroup far away (another XCD)
        const unsigned v = __builtin_nontemporal_load(&data[(size_t)(b & 1) * G * 256 + o * 256 + threadIdx.x]);
        if (v != (unsigned)(b * 1000003 + o * 256 + threadIdx.x)) atomicAdd(bad, 1u);
        // a second barrier per round would be needed before the slot is rewritten two rounds later: (b & 1) alternation + the next barrier covers it
    }
}

int main() {
    unsigned *counter, *data, *bad; int* fail;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&fail, 4)); CK(hipMalloc(&data, 2 * 512 * 256 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int G : {16, 64, 128, 256}) {
        for (int nb : {1, 11, 101}) {
            float best = 1e9f;
            unsigned hbad = 0; int hfail = 0;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipMemsetAsync(counter, 0, 4, s)); CK(hipMemsetAsync(bad, 0, 4, s)); CK(hipMemsetAsync(fail, 0, 4, s));
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, s, counter, data, nb, fail, bad);
                CK(hipEventRecord(e1, s));
                CK(hipStreamSynchronize(s));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hfail, fail, 4, hipMemcpyDeviceToHost));
                if (hbad || hfail) break;
            }
            printf("G=%3d barriers=%3d: %.2f us per launch (best of 6)  stale reads %u  spin limit hit %d\n", G, nb, best * 1e3f, hbad, hfail);
        }
    }
    return 0;
}
