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

// test aid: leaves quiet-NaN bit patterns in the whole LDS of every CU.  LDS is not cleared between kernels, so a kernel that reads a word
// it (or its predecessor in the same launch) never wrote sees whatever ran before it -- usually harmless values, which is how such a read
// survives testing (round 2: the padding behind K2's dustbin entry).  Running this first makes the stale data poisonous.
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned* sink, int words) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = 0x7fc00000u;
    __syncthreads();
    if (sink && lds[(threadIdx.x * 37) % words] != 0x7fc00000u) sink[0] = 1u;      // keeps the stores alive
}

}  // namespace s2m2

extern "C" int s2m2_debug_poison_lds(void* stream) {
    using namespace s2m2;
    constexpr int kBytes = 160 * 1024;                            // one block owns a CU's whole LDS; 4 blocks per CU's worth of grid
    static bool attr_done_dev[kMaxDevices] = {};
    bool& attr_done = attr_done_dev[current_device()];
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBytes) != hipSuccess)
            return set_error("debug_poison_lds: cannot reserve %d bytes of LDS", kBytes);
        attr_done = true;
    }
    hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), kBytes, static_cast<hipStream_t>(stream), (unsigned*)nullptr, kBytes / 4);
    return check_launch("debug_poison_lds");
}

extern "C" int s2m2_version(void) { return 100; }          // 0.1.0
extern "C" const char* s2m2_last_error(void) { return s2m2::g_err; }
