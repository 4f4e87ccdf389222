// ABI-level plumbing shared by every entry point of libance_amd.so.
#include "common.h"
#include <string.h>

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
}  // namespace ance

extern "C" int ance_abi_version(void) { return ANCE_ABI_VERSION; }
extern "C" const char *ance_last_error(void) { return ance::g_err; }
