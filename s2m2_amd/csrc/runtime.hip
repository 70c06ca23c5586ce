// Host-side runtime glue of libs2m2_hip.so: version, thread-local error text, launch checks.
#include "common.h"

namespace s2m2 {

static thread_local char g_err[512] = "";

int set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

}  // namespace s2m2

extern "C" int s2m2_version(void) { return 100; }          // 0.1.0
extern "C" const char* s2m2_last_error(void) { return s2m2::g_err; }
