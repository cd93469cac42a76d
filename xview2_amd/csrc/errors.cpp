// Thread-local error string of the C ABI (include/xv2.h: xv2_last_error).
#include <stdarg.h>
#include <stdio.h>
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/xv2.h"
#include "amax_ctx.h"

namespace xv2 {
AmaxCtx& amax_ctx() {
    static thread_local AmaxCtx c;
    return c;
}
int& amax_depth() {
    static thread_local int d = 0;
    return d;
}
}  // namespace xv2

extern "C" int xv2_amax_ctx(const void* amax_a0, const void* amax_a1, const void* amax_dy, void* amax_out) {
    xv2::AmaxCtx& c = xv2::amax_ctx();
    c.a0 = static_cast<const unsigned*>(amax_a0);
    c.a1 = static_cast<const unsigned*>(amax_a1);
    c.dy = static_cast<const unsigned*>(amax_dy);
    c.out = static_cast<unsigned*>(amax_out);
    return 0;
}
namespace xv2 {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace xv2
extern "C" const char* xv2_last_error(void) { return xv2::g_err; }
extern "C" int xv2_version(void) { return 1; }

// ---- profiler -------------------------------------------------------------------------------
namespace xv2 {
struct ProfRec { int kid; double flops, bytes; hipEvent_t a, b; };
static bool g_prof_on = false;
static int g_prof_only = -1;
static bool g_prof_open = false;
static int g_prof_stride = 1;   // bracket every n-th eligible launch
static long g_prof_seen = 0;   // >= 0: bracket only launches of this kernel id
static std::vector<std::string> g_prof_names;
static std::vector<ProfRec> g_prof_recs;
static std::vector<hipEvent_t> g_prof_pool;
static size_t g_prof_pool_next = 0;
static hipEvent_t prof_event() {
    if (g_prof_pool_next == g_prof_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        g_prof_pool.push_back(e);
    }
    return g_prof_pool[g_prof_pool_next++];
}
static std::mutex g_prof_mu;
int prof_register(const char* name) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_names.emplace_back(name);
    return (int)g_prof_names.size() - 1;
}
void prof_begin(int kid, double flops, double bytes, hipStream_t stream) {
    if (!g_prof_on || (g_prof_only >= 0 && kid != g_prof_only) || (g_prof_seen++ % g_prof_stride) != 0) {
        g_prof_open = false;
        return;
    }
    g_prof_open = true;
    ProfRec r{kid, flops, bytes, prof_event(), prof_event()};
    (void)hipEventRecord(r.a, stream);
    g_prof_recs.push_back(r);
}
void prof_end(hipStream_t stream) {
    if (!g_prof_on || !g_prof_open || g_prof_recs.empty()) return;
    g_prof_open = false;
    (void)hipEventRecord(g_prof_recs.back().b, stream);
}
}  // namespace xv2
extern "C" int xv2_prof_enable(int on) {
    xv2::g_prof_on = on != 0;
    xv2::g_prof_only = on >= 2 ? on - 2 : -1;
    if (on) {
        xv2::g_prof_recs.clear();
        xv2::g_prof_pool_next = 0;
    }
    return XV2_OK;
}
extern "C" int xv2_prof_stride(int n) {
    xv2::g_prof_stride = n > 1 ? n : 1;
    xv2::g_prof_seen = 0;
    return XV2_OK;
}
extern "C" int xv2_prof_num_kernels(void) { return (int)xv2::g_prof_names.size(); }
extern "C" const char* xv2_prof_kernel_name(int kid) {
    return (kid >= 0 && kid < (int)xv2::g_prof_names.size()) ? xv2::g_prof_names[kid].c_str() : "";
}
extern "C" int xv2_prof_summary(int kid, double* total_ms, double* total_flops, double* total_bytes,
                                int64_t* launches) {
    double ms = 0.0, fl = 0.0, by = 0.0;
    int64_t n = 0;
    for (auto& r : xv2::g_prof_recs) {
        if (r.kid != kid) continue;
        if (hipEventSynchronize(r.b) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
        ms += t;
        fl += r.flops;
        by += r.bytes;
        ++n;
    }
    *total_ms = ms;
    *total_flops = fl;
    *total_bytes = by;
    *launches = n;
    return XV2_OK;
}
extern "C" int xv2_prof_num_records(void) { return (int)xv2::g_prof_recs.size(); }
extern "C" int xv2_prof_record(int i, int* kid, double* ms, double* flops, double* bytes) {
    if (i < 0 || i >= (int)xv2::g_prof_recs.size()) return XV2_EINVAL;
    auto& r = xv2::g_prof_recs[i];
    float t = 0.f;
    if (hipEventSynchronize(r.b) != hipSuccess || hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return XV2_EHIP;
    *kid = r.kid;
    *ms = t;
    *flops = r.flops;
    *bytes = r.bytes;
    return XV2_OK;
}
