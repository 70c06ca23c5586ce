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

int current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d < kMaxDevices ? d : kMaxDevices - 1;
}

// 256 zero bytes per device (source of the padding / tail pieces of the GEMM loaders), allocated on the first call on that device
// -- before any graph capture: the engine warms up eagerly
const void* zero_page() {
    static void* z[kMaxDevices] = {};
    void*& p = z[current_device()];
    if (!p) {
        if (hipMalloc(&p, 256) != hipSuccess || hipMemset(p, 0, 256) != hipSuccess) p = nullptr;
    }
    return p;
}

}  // namespace s2m2

extern "C" int s2m2_version(void) { return 100; }          // 0.1.0
extern "C" const char* s2m2_last_error(void) { return s2m2::g_err; }
