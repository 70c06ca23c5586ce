// Host-side runtime glue of libs2m2_hip.so: version, thread-local error text, launch checks.
#include "common.h"

#include <mutex>

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
static std::mutex& host_mutex() {
    static std::mutex m;
    return m;
}

int reserve_lds(const void* func, size_t bytes, size_t* granted, const char* what) {
    std::lock_guard<std::mutex> lock(host_mutex());
    size_t& g = granted[current_device()];
    if (bytes > g) {
        if (hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess)
            return set_error("%s: cannot reserve %zu bytes of LDS", what, bytes);
        g = bytes;
    }
    return 0;
}

const void* zero_page() {
    static void* z[kMaxDevices] = {};
    std::lock_guard<std::mutex> lock(host_mutex());
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

// measurement aid: one lane stamps the shader-clock counter (s_memtime: ticks at the CU's CURRENT engine clock) and the constant 100 MHz
// real-time counter (s_memrealtime).  Two probes on one stream around a region give the average engine clock the region ran at --
// how tools/clock_probe.py tells a clock-limited (power-managed) in-forward kernel time from an idle-chip micro-benchmark.
__global__ void clock_probe_kernel(unsigned long long* out) {
    if (threadIdx.x == 0) {
        out[0] = __builtin_amdgcn_s_memtime();
        out[1] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace s2m2

extern "C" int s2m2_debug_clock_probe(void* out, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(out, "debug_clock_probe: null pointer");
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), static_cast<unsigned long long*>(out));
    return check_launch("debug_clock_probe");
}

extern "C" int s2m2_debug_poison_lds(void* stream) {
    using namespace s2m2;
    constexpr int kBytes = 160 * 1024;                            // one block owns a CU's whole LDS; 4 blocks per CU's worth of grid
    static size_t granted[kMaxDevices] = {};
    if (reserve_lds(reinterpret_cast<const void*>(poison_lds_kernel), kBytes, granted, "debug_poison_lds")) return 1;
    hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), kBytes, static_cast<hipStream_t>(stream), (unsigned*)nullptr, kBytes / 4);
    return check_launch("debug_poison_lds");
}

extern "C" int s2m2_version(void) { return S2M2_ABI_VERSION; }   // include/s2m2_hip.h
extern "C" const char* s2m2_last_error(void) { return s2m2::g_err; }
