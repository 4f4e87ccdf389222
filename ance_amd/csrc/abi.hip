// ABI-level plumbing shared by every entry point of libance_amd.so.
#include "common.h"
#include <string.h>
#include <vector>

namespace ance {
static thread_local char g_err[256] = "";

void set_last_error(const char *msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return ANCE_OK;
    char buf[256];
    snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
    set_last_error(buf);
    return ANCE_E_LAUNCH;
}

// ---- profiler ------------------------------------------------------------------------------
namespace {
struct ProfRec { int cat; hipEvent_t a, b; double work; };
bool g_prof_on = false;
std::vector<ProfRec> g_recs;
double g_ms[PC_COUNT], g_work[PC_COUNT];
long long g_cnt[PC_COUNT];
void prof_drain() {
    for (auto &r : g_recs) {
        float ms = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_ms[r.cat] += ms;
            g_work[r.cat] += r.work;
            g_cnt[r.cat] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_recs.clear();
}
}  // namespace
bool prof_enabled() { return g_prof_on; }
void prof_begin(int cat, hipStream_t st, double work) {
    ProfRec r;
    r.cat = cat;
    r.work = work;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
}
void prof_end(hipStream_t st) {
    if (!g_recs.empty()) (void)hipEventRecord(g_recs.back().b, st);
    if (g_recs.size() >= 4096) prof_drain();
}
}  // namespace ance

extern "C" void ance_profile_enable(int on) {
    using namespace ance;
    prof_drain();
    g_prof_on = on != 0;
    for (int i = 0; i < PC_COUNT; ++i) { g_ms[i] = 0; g_work[i] = 0; g_cnt[i] = 0; }
}
extern "C" int ance_profile_read(double *ms, double *work, long long *count, int n) {
    using namespace ance;
    prof_drain();
    for (int i = 0; i < n && i < PC_COUNT; ++i) { ms[i] = g_ms[i]; work[i] = g_work[i]; count[i] = g_cnt[i]; }
    return PC_COUNT;
}

extern "C" int ance_abi_version(void) { return ANCE_ABI_VERSION; }
extern "C" const char *ance_last_error(void) { return ance::g_err; }
