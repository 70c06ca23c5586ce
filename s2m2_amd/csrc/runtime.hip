// Host-side runtime glue of libs2m2_hip.so: version, thread-local error text, launch checks.
#include "common.h"
#include "plan.h"

#include <mutex>
#include <string.h>
#include <vector>

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

// ------------------------------------------------------------------------------------------------------------------------------------
// Recorded launch plans (include/s2m2_hip.h: s2m2_plan_*).  A plan is the list of library calls one thread made between s2m2_plan_begin and
// s2m2_plan_end -- each as a flat argument blob plus a trampoline (plan.h) -- with the pointers into the caller's EXTERNAL buffers (declared at
// s2m2_plan_end) stored relative to their buffer, so that s2m2_plan_run can re-issue the whole sequence from C++ with the externals somewhere
// else.  Every other pointer (weights, the scratch and intermediate tensors of the recorded run) is replayed as recorded: whoever records keeps
// those allocations alive for the life of the plan.  Only the POINTER words of a blob (plan.h: PlanPtrMask -- from the argument types of a
// positional pack, from the pointer-field list of a descriptor) are compared with the external ranges: a size, a stride or a pair of ints that
// happens to fall inside a range is left alone.
// ------------------------------------------------------------------------------------------------------------------------------------
struct s2m2_plan {
    struct Call { int (*tramp)(const void*, void*); size_t off, words; const char* name; s2m2::PlanPtrMask mask; };
    struct Patch { int call; int word; int slot; long long delta; };
    std::vector<unsigned long long> arena;      // the blobs, 8-byte aligned
    std::vector<Call> calls;
    std::vector<Patch> patches;
    int nslots = 0;
    bool sealed = false, failed = false;
    size_t max_words = 0;
};

namespace s2m2 {
static thread_local s2m2_plan* g_plan = nullptr;

bool plan_recording() { return g_plan != nullptr; }

int plan_append(int (*tramp)(const void*, void*), const void* blob, size_t bytes, const char* name, const PlanPtrMask& mask) {
    s2m2_plan* p = g_plan;
    if (!p) return 0;
    const size_t words = (bytes + 7) / 8;
    if (mask.overflow || words > 64 * (size_t)kPlanMaskWords) {
        p->failed = true;
        return set_error("plan: %s has a pointer argument outside the %d words the pointer mask covers (or a misaligned one)", name, 64 * kPlanMaskWords);
    }
    const size_t off = p->arena.size();
    p->arena.resize(off + words, 0ULL);
    memcpy(p->arena.data() + off, blob, bytes);
    p->calls.push_back({tramp, off, words, name, mask});
    if (words > p->max_words) p->max_words = words;
    return 0;
}
}  // namespace s2m2

extern "C" int s2m2_plan_begin(s2m2_plan** plan) {
    using namespace s2m2;
    S2M2_REQUIRE(plan, "plan_begin: null pointer");
    S2M2_REQUIRE(!g_plan, "plan_begin: this thread is recording a plan already");
    *plan = new s2m2_plan();
    g_plan = *plan;
    return 0;
}

extern "C" int s2m2_plan_end(s2m2_plan* plan, const void* const* ext_base, const size_t* ext_bytes, int next) {
    using namespace s2m2;
    S2M2_REQUIRE(plan && g_plan == plan, "plan_end: not the plan this thread is recording");
    g_plan = nullptr;
    S2M2_REQUIRE(next >= 0 && next <= 16 && (next == 0 || (ext_base && ext_bytes)), "plan_end: next=%d external buffers (0..16)", next);
    plan->nslots = next;
    for (int c = 0; c < (int)plan->calls.size(); ++c) {
        const auto& call = plan->calls[c];
        for (size_t wd = 1; wd < call.words; ++wd) {               // word 0 is the function pointer of the call
            if (!call.mask.test(wd)) continue;                     // not a pointer argument / field
            const unsigned long long v = plan->arena[call.off + wd];
            for (int s = 0; s < next; ++s) {
                const unsigned long long b = (unsigned long long)(uintptr_t)ext_base[s];
                if (b && ext_bytes[s] && v >= b && v < b + ext_bytes[s]) {
                    plan->patches.push_back({c, (int)wd, s, (long long)(v - b)});
                    break;
                }
            }
        }
    }
    plan->sealed = true;
    return 0;
}

extern "C" int s2m2_plan_abort(s2m2_plan* plan) {                    // stop recording without a usable plan (an exception between begin and end)
    if (s2m2::g_plan == plan) s2m2::g_plan = nullptr;
    if (plan) plan->failed = true;
    return 0;
}

extern "C" int s2m2_plan_launches(const s2m2_plan* plan) { return plan ? (int)plan->calls.size() : -1; }

// recorded pointers that follow external buffer `slot` (diagnostics; slot < 0: all)
extern "C" int s2m2_plan_patches(const s2m2_plan* plan, int slot) {
    if (!plan) return -1;
    int n = 0;
    for (const auto& pt : plan->patches) n += slot < 0 || pt.slot == slot;
    return n;
}

extern "C" int s2m2_plan_run(const s2m2_plan* plan, const void* const* ext_ptrs, int next, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(plan && plan->sealed && !plan->failed, "plan_run: the plan was not recorded to its end");
    S2M2_REQUIRE(next == plan->nslots && (next == 0 || ext_ptrs), "plan_run: %d external buffers given, the plan was recorded with %d", next, plan->nslots);
    S2M2_REQUIRE(!g_plan, "plan_run: this thread is recording a plan");
    std::vector<unsigned long long> blob(plan->max_words);           // a private copy per call: concurrent runs of one plan do not share state
    size_t pi = 0;
    for (int c = 0; c < (int)plan->calls.size(); ++c) {
        const auto& call = plan->calls[c];
        memcpy(blob.data(), plan->arena.data() + call.off, call.words * 8);
        for (; pi < plan->patches.size() && plan->patches[pi].call == c; ++pi) {
            const auto& pt = plan->patches[pi];
            S2M2_REQUIRE(ext_ptrs[pt.slot], "plan_run: external buffer %d is null but call %d (%s) uses it", pt.slot, c, call.name);
            blob[pt.word] = (unsigned long long)(uintptr_t)ext_ptrs[pt.slot] + (unsigned long long)pt.delta;
        }
        if (call.tramp(blob.data(), stream) != 0) return 1;           // (the entry point's own message is in s2m2_last_error)
    }
    return 0;
}

extern "C" int s2m2_plan_destroy(s2m2_plan* plan) {
    if (s2m2::g_plan == plan) s2m2::g_plan = nullptr;
    delete plan;
    return 0;
}

// One refinement iteration as ONE native call (LocalRefiner.forward + the loop epilogue, refinenet.py:126-154, s2m2.py:175-180: K3, the
// corr / disparity / confidence feature layers, the U-Net with its attention blocks, the ConvGRU, the update heads and refine_update --
// about 55 launches): a plan recorded around that iteration whose externals are, in this order, the iteration's inputs.
extern "C" int s2m2_refine_step(const s2m2_plan* step, const void* hidden, const void* ctx, const void* disp, const void* conf, const void* occ,
                                const void* cv, const void* side_input, void* stream) {
    const void* ext[7] = {hidden, ctx, disp, conf, occ, cv, side_input};
    return s2m2_plan_run(step, ext, 7, stream);
}

#if S2M2_RANGE_CHECK
// ---- fp16 headroom probe (common.h): the range word, the per-launch log and its read-out (libs2m2_hip_range.so only) --------------------
namespace s2m2 {
constexpr int kRangeLog = 4096;
static unsigned* g_range_dev[kMaxDevices] = {};           // [0]: the live word, [1 .. kRangeLog]: the log
static std::vector<const char*> g_range_names;
static std::mutex g_range_mutex;

unsigned* range_word() {
    std::lock_guard<std::mutex> lock(g_range_mutex);
    unsigned*& p = g_range_dev[current_device()];
    if (!p && (hipMalloc(&p, (kRangeLog + 1) * sizeof(unsigned)) != hipSuccess || hipMemset(p, 0, (kRangeLog + 1) * sizeof(unsigned)) != hipSuccess)) p = nullptr;
    return p;
}

__global__ void range_collect_kernel(unsigned* buf, int slot) {
    buf[1 + slot] = buf[0];
    buf[0] = 0u;
}

int range_collect(const char* name, void* stream) {
    unsigned* buf = range_word();
    if (!buf) return set_error("range probe: cannot allocate the log");
    int slot;
    {
        std::lock_guard<std::mutex> lock(g_range_mutex);
        slot = (int)g_range_names.size();
        if (slot >= kRangeLog) return 0;                           // log full: later launches are not recorded
        g_range_names.push_back(name);
    }
    hipLaunchKernelGGL(range_collect_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), buf, slot);
    return check_launch("range_collect");
}
}  // namespace s2m2

// read-out: synchronises the device; returns the number of logged launches, fills max |value| converted to fp16 per launch (up to cap);
// s2m2_debug_range_name(i): the entry point of launch i; s2m2_debug_range_reset(): empties the log
extern "C" int s2m2_debug_range_log(float* maxima, int cap) {
    using namespace s2m2;
    unsigned* buf = range_word();
    if (!buf || hipDeviceSynchronize() != hipSuccess) return -1;
    const int n = (int)g_range_names.size();
    std::vector<unsigned> host((size_t)n + 1);
    if (n && hipMemcpy(host.data(), buf, ((size_t)n + 1) * sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    for (int i = 0; i < n && i < cap; ++i) memcpy(&maxima[i], &host[1 + i], sizeof(float));
    return n;
}
extern "C" const char* s2m2_debug_range_name(int i) {
    return (i >= 0 && i < (int)s2m2::g_range_names.size()) ? s2m2::g_range_names[i] : "";
}
extern "C" int s2m2_debug_range_reset(void) {
    std::lock_guard<std::mutex> lock(s2m2::g_range_mutex);
    s2m2::g_range_names.clear();
    return 0;
}
#endif

extern "C" int s2m2_version(void) { return S2M2_ABI_VERSION; }   // include/s2m2_hip.h
extern "C" const char* s2m2_last_error(void) { return s2m2::g_err; }
