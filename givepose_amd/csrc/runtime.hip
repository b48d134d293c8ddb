// Error reporting, device info, hipGraph capture and per-launch event timing.
#include <stdarg.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "common.hpp"

thread_local char gp_err_buf[512] = {0};

int gp_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(gp_err_buf, sizeof(gp_err_buf), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* gp_last_error(void) { return gp_err_buf; }
extern "C" int gp_version(void) { return GP_ABI_VERSION; }

extern "C" int gp_device_info(int* cu_count, char* arch, int arch_len) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return gp_fail(GP_ERR_RUNTIME, "hipGetDevice failed");
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, dev) != hipSuccess) return gp_fail(GP_ERR_RUNTIME, "hipGetDeviceProperties failed");
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (arch && arch_len > 0) {
        strncpy(arch, p.gcnArchName, arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    return GP_OK;
}

// ------------------------------------------------------------------------------------ graphs
extern "C" int gp_graph_begin(void* stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return gp_fail(GP_ERR_RUNTIME, "hipStreamBeginCapture: %s", hipGetErrorString(e));
    return GP_OK;
}
extern "C" int gp_graph_end(void* stream, void** graph_exec_out) {
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess) return gp_fail(GP_ERR_RUNTIME, "hipStreamEndCapture: %s", hipGetErrorString(e));
    hipGraphExec_t ge = nullptr;
    e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) return gp_fail(GP_ERR_RUNTIME, "hipGraphInstantiate: %s", hipGetErrorString(e));
    *graph_exec_out = (void*)ge;
    return GP_OK;
}
extern "C" int gp_graph_launch(void* graph_exec, void* stream) {
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) return gp_fail(GP_ERR_RUNTIME, "hipGraphLaunch: %s", hipGetErrorString(e));
    return GP_OK;
}
extern "C" int gp_graph_destroy(void* graph_exec) {
    if (graph_exec) hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return GP_OK;
}

// ------------------------------------------------------------------------------------ timing
namespace {
struct Rec {
    hipEvent_t a, b;
    int cls;
    double flops, bytes;
    const char* name;
    std::string label;
};
bool g_timing = false;
hipStream_t g_tstream = nullptr;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
Rec g_cur;
bool g_open = false;
struct Acc {
    long n = 0;
    double ms = 0, flops = 0, bytes = 0;
    int cls = 0;
} g_acc[GP_KC_COUNT];
std::vector<std::pair<std::string, Acc>> g_top;

hipEvent_t get_event() {
    if (!g_pool.empty()) {
        hipEvent_t e = g_pool.back();
        g_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    hipEventCreate(&e);
    return e;
}
}  // namespace

void gp_timing_before(hipStream_t s, int cls, double flops, double bytes) {
    if (!g_timing || s != g_tstream) return;
    g_cur.a = get_event();
    g_cur.b = get_event();
    g_cur.cls = cls;
    g_cur.flops = flops;
    g_cur.bytes = bytes;
    g_cur.label.clear();
    hipEventRecord(g_cur.a, s);
    g_open = true;
}

void gp_timing_label(const char* fmt, ...) {
    if (!g_open) return;
    char buf[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_cur.label = buf;
}

int gp_timing_after(const char* name) {
    if (g_open) {
        g_cur.name = name;
        hipEventRecord(g_cur.b, g_tstream);
        g_recs.push_back(g_cur);
        g_open = false;
    }
    return GP_OK;
}

extern "C" int gp_timing_begin(void* stream) {
    g_tstream = (hipStream_t)stream;
    g_recs.clear();
    for (auto& a : g_acc) a = Acc();
    g_top.clear();
    g_timing = true;
    return GP_OK;
}

extern "C" int gp_timing_end(void) {
    g_timing = false;
    hipError_t e = hipStreamSynchronize(g_tstream);
    if (e != hipSuccess) return gp_fail(GP_ERR_RUNTIME, "timing sync: %s", hipGetErrorString(e));
    // GP_TIMING_DUMP=<file>: one line per launch (sequence number, class, entry point, us, algorithmic FLOP / bytes)
    const char* dump = getenv("GP_TIMING_DUMP");
    FILE* df = dump ? fopen(dump, "w") : nullptr;
    int seq = 0;
    std::map<std::string, Acc> by_label;
    for (auto& r : g_recs) {
        float ms = 0;
        hipEventElapsedTime(&ms, r.a, r.b);
        if (df) fprintf(df, "%d %d %s %.2f %.0f %.0f %s\n", seq++, r.cls, r.name ? r.name : "?", ms * 1e3, r.flops, r.bytes, r.label.c_str());
        Acc& a = g_acc[r.cls];
        a.n++;
        a.ms += ms;
        a.flops += r.flops;
        a.bytes += r.bytes;
        Acc& t = by_label[r.label.empty() ? std::string(r.name ? r.name : "?") : r.label];
        t.n++;
        t.ms += ms;
        t.flops += r.flops;
        t.bytes += r.bytes;
        t.cls = r.cls;
        g_pool.push_back(r.a);
        g_pool.push_back(r.b);
    }
    if (df) fclose(df);
    g_recs.clear();
    g_top.assign(by_label.begin(), by_label.end());
    std::sort(g_top.begin(), g_top.end(), [](const auto& x, const auto& y) { return x.second.ms > y.second.ms; });
    return GP_OK;
}

extern "C" int gp_timing_report(int cls, long* launches, double* ms, double* flops, double* bytes) {
    if (cls < 0 || cls >= GP_KC_COUNT) return gp_fail(GP_ERR_INVALID, "bad kernel class %d", cls);
    *launches = g_acc[cls].n;
    *ms = g_acc[cls].ms;
    *flops = g_acc[cls].flops;
    *bytes = g_acc[cls].bytes;
    return GP_OK;
}

extern "C" int gp_timing_top(int rank, char* label, int label_len, int* cls, long* launches, double* ms, double* flops, double* bytes) {
    if (rank < 0 || rank >= (int)g_top.size()) return gp_fail(GP_ERR_INVALID, "gp_timing_top: rank %d of %d", rank, (int)g_top.size());
    const auto& e = g_top[rank];
    if (label && label_len > 0) {
        strncpy(label, e.first.c_str(), label_len - 1);
        label[label_len - 1] = 0;
    }
    if (cls) *cls = e.second.cls;
    *launches = e.second.n; *ms = e.second.ms; *flops = e.second.flops; *bytes = e.second.bytes;
    return GP_OK;
}
