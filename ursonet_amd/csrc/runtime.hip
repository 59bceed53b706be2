// Host-side runtime glue of liburso_hip.so: error reporting and the opt-in launch profiler.
#include "common.h"
#include <stdarg.h>
#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void urso_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int urso_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { urso_set_error("%s: %s", what, hipGetErrorString(e)); return URSO_ELAUNCH; }
    return URSO_OK;
}

extern "C" const char* urso_last_error(void) { return g_err; }
extern "C" int urso_abi_version(void) { return 9; }

// ---------------------------------------------------------------- explicit policy options
UrsoOptions g_urso_opt;
#include <string.h>
static int* opt_slot(const char* name) {
    if (!name) return nullptr;
#define URSO_OPT(n) if (!strcmp(name, #n)) return &g_urso_opt.n;
    URSO_OPT(pw_kernel) URSO_OPT(pw_small) URSO_OPT(igemm_shortk) URSO_OPT(wgrad_narrow) URSO_OPT(wgrad_blocks) URSO_OPT(wgrad_pipe) URSO_OPT(wgrad_ring) URSO_OPT(wgrad_big)
    URSO_OPT(grid_cap) URSO_OPT(hconv) URSO_OPT(hconv_dbg) URSO_OPT(hconv2) URSO_OPT(hconv2_shape) URSO_OPT(hconv_streamk) URSO_OPT(pair) URSO_OPT(pair_single) URSO_OPT(c3) URSO_OPT(c3v) URSO_OPT(stem) URSO_OPT(stem_pool) URSO_OPT(cus) URSO_OPT(hwgrad) URSO_OPT(dense) URSO_OPT(pwx) URSO_OPT(pwx_bn) URSO_OPT(pwx_dbg) URSO_OPT(bneck) URSO_OPT(mold_scalar)
#undef URSO_OPT
    return nullptr;
}
int urso_device_cus() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0; hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ncu = pr.multiProcessorCount;
        if (ncu <= 0) ncu = 256;
    }
    return ncu;
}
int urso_usable_cus() {
    const int hw = urso_device_cus(), want = g_urso_opt.cus;
    if (want <= 0 || want >= hw) return hw;
    return want < 8 ? 8 : want / 8 * 8;
}
extern "C" int urso_set_option(const char* name, int value) {
    int* s = opt_slot(name);
    if (!s) { urso_set_error("urso_set_option: unknown option '%s'", name ? name : "(null)"); return URSO_EINVAL; }
    if (!strcmp(name, "wgrad_blocks") && value < 1) { urso_set_error("urso_set_option: wgrad_blocks must be >= 1"); return URSO_EINVAL; }
    *s = value;
    return URSO_OK;
}
extern "C" int urso_get_option(const char* name, int* value) {
    int* s = opt_slot(name);
    if (!s || !value) { urso_set_error("urso_get_option: unknown option '%s'", name ? name : "(null)"); return URSO_EINVAL; }
    *value = *s;
    return URSO_OK;
}

// ---------------------------------------------------------------- profiler
struct ProfRec { int id; double flops, bytes, l2; hipEvent_t e0, e1; const void* fn; hipStream_t st; int nl; bool open; };
static std::mutex g_pmu;
static bool g_prof_on = false;
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}

static thread_local int g_prof_depth = 0;      // nested entry points (urso_conv_winograd_fwd -> urso_conv_igemm_ex): only the outermost call is a record

void urso_prof_before(hipStream_t s, int kernel_id, double flops, double bytes) {
    if (g_prof_depth++ > 0) return;
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_pmu);
    ProfRec r; r.id = kernel_id; r.flops = flops; r.bytes = bytes; r.l2 = 0.0; r.e0 = get_event(); r.e1 = get_event(); r.fn = nullptr; r.st = s; r.nl = 0; r.open = true;
    (void)hipEventRecord(r.e0, s);
    g_recs.push_back(r);
}
void urso_prof_after(hipStream_t s) {
    if (--g_prof_depth > 0) return;
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_pmu);
    if (!g_recs.empty()) { (void)hipEventRecord(g_recs.back().e1, s); g_recs.back().open = false; }
}
// bytes the launch copies from L2 into LDS / registers (tile re-reads included): the third roof next to HBM and the matrix pipe
void urso_prof_l2(double l2_bytes) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_pmu);
    if (g_recs.empty() || !g_recs.back().open) return;
    g_recs.back().l2 += l2_bytes;
}
void urso_prof_symbol(const void* host_fn) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_pmu);
    if (g_recs.empty() || !g_recs.back().open) return;           // a launch outside any profiled entry point
    ProfRec& r = g_recs.back();
    if (!r.fn) r.fn = host_fn;                                    // the first kernel of the call names it (a split-K finish or a reduction follows it)
    ++r.nl;
}

extern "C" int urso_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_pmu);
    g_prof_on = on != 0;
    return URSO_OK;
}

extern "C" int urso_prof_collect_ex(urso_prof_record_ex* out, int max_records) {
    std::lock_guard<std::mutex> lk(g_pmu);
    int n = 0;
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        if (out && n < max_records) {
            urso_prof_record_ex& o = out[n];
            o.kernel_id = r.id; o.ms = ms; o.flops = r.flops; o.bytes = r.bytes; o.l2_bytes = r.l2; o.n_launches = r.nl;
            const char* nm = r.fn ? hipKernelNameRefByPtr(r.fn, r.st) : nullptr;
            snprintf(o.symbol, sizeof(o.symbol), "%s", nm ? nm : "");
            ++n;
        }
        g_pool.push_back(r.e0); g_pool.push_back(r.e1);
    }
    g_recs.clear();
    return n;
}

extern "C" int urso_prof_collect(urso_prof_record* out, int max_records) {
    std::lock_guard<std::mutex> lk(g_pmu);
    int n = 0;
    for (auto& r : g_recs) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        if (out && n < max_records) { out[n].kernel_id = r.id; out[n].ms = ms; out[n].flops = r.flops; out[n].bytes = r.bytes; ++n; }
        g_pool.push_back(r.e0); g_pool.push_back(r.e1);
    }
    g_recs.clear();
    return n;
}
