// Shared device/host helpers for the S2M2 gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/s2m2_hip.h"

namespace s2m2 {

// ------------------------------------------------------------------------------------------------
// host: error reporting
// ------------------------------------------------------------------------------------------------
int set_error(const char* fmt, ...);            // stores a thread-local message, returns 1
int check_launch(const char* what);             // hipGetLastError() -> set_error
// Per-device host state.  A process may drive several GPUs (load_model(..., device='cuda:1') next to cuda:0): everything the
// library caches on the host is indexed by the calling thread's current HIP device (the Python binding makes the tensors'
// device current around every call).
constexpr int kMaxDevices = 32;
int current_device();                           // hipGetDevice(), clamped to [0, kMaxDevices)
const void* zero_page();                        // 256 zero bytes in the current device's memory (allocated on first use, never freed)
// Raises the dynamic-LDS limit of kernel `func` on the current device to at least `bytes` (hipFuncSetAttribute), once per size:
// `granted` is the caller's per-instantiation cache (a zero-initialised static array of kMaxDevices entries).  Serialised by one
// process-wide mutex: forward() may be called from any thread, and two modules on the same device share these caches.
int reserve_lds(const void* func, size_t bytes, size_t* granted, const char* what);

#define S2M2_REQUIRE(cond, ...)                        \
    do {                                               \
        if (!(cond)) return ::s2m2::set_error(__VA_ARGS__); \
    } while (0)

// ------------------------------------------------------------------------------------------------
// device: types
// ------------------------------------------------------------------------------------------------
typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <typename T> struct DT;
template <> struct DT<float>  { static constexpr int code = S2M2_F32; static constexpr int vec = 4; };
template <> struct DT<half_t> { static constexpr int code = S2M2_F16; static constexpr int vec = 8; };

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(half_t x) { return (float)x; }
// fp16 headroom probe (experiment builds only: -DS2M2_RANGE_CHECK=1 -> libs2m2_hip_range.so, tools/range_report.py): every fp32 -> fp16
// conversion of the library goes through from_f32<half_t>; in the probe build it also folds |x| into ONE device word (atomic max on the bit
// pattern: non-negative floats order like unsigned integers, NaN patterns sort above infinity), which the host moves into a per-launch log
// after every launch (plan.h: plan_dispatch).  A trained checkpoint whose activations leave the fp16 range shows up as a ratio >= 1 (or NaN)
// against 65504 in the layer that produces them -- before the value is rounded to inf.  The shipped library carries none of this.
#ifndef S2M2_RANGE_CHECK
#define S2M2_RANGE_CHECK 0
#endif
#if S2M2_RANGE_CHECK
static __device__ unsigned* g_range_word_tu = nullptr;    // this translation unit's copy of the pointer to the library's range word
unsigned* range_word();                                   // runtime.hip: the word (device memory, per device), allocated on first use
int range_collect(const char* name, void* stream);        // runtime.hip: log[n++] = {name, word}; word = 0   (one tiny launch)
static void range_bind_tu() {                             // (internal linkage: one copy, and one `bound` flag, PER translation unit) called by plan_dispatch before every launch
    static bool bound[kMaxDevices] = {};
    const int dev = current_device();
    if (!bound[dev]) {
        unsigned* w = range_word();
        if (w && hipMemcpyToSymbol(HIP_SYMBOL(g_range_word_tu), &w, sizeof(w)) == hipSuccess) bound[dev] = true;
    }
}
__device__ __forceinline__ void range_note(float x) {
    unsigned* w = g_range_word_tu;
    const unsigned b = __builtin_bit_cast(unsigned, x) & 0x7fffffffu;
    if (w && b > *reinterpret_cast<volatile unsigned*>(w)) atomicMax(w, b);
    // the kernels with untracked load rings count their waits (wait_vmcnt<N>): an atomic in flight ticks the same counter and may retire out of
    // order with the loads, which would let a counted wait pass early -- drain before going on (the probe build is not a fast build)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#endif

template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ half_t from_f32<half_t>(float x) {
#if S2M2_RANGE_CHECK
    range_note(x);
#endif
    return (half_t)x;
}

// load through an explicit global (address space 1) pointer: keeps the access a global_load even where the compiler cannot
// prove the address space of a selected pointer (a flat load would also tick lgkmcnt and serialise with the LDS traffic)
typedef float raw16_t __attribute__((ext_vector_type(4)));       // 16 opaque bytes in 4 VGPRs
__device__ __forceinline__ raw16_t global_load16(const void* p) {
    return *(const __attribute__((address_space(1))) raw16_t*)(p);
}

// Asynchronous 16-byte global load the compiler does NOT track: no automatic s_waitcnt.  The compiler's own vmcnt bookkeeping is
// conservative across loop back edges and uniform branches -- in a register ring of loads it waits for the previous iteration's
// load before it issues the next one, which turns an N-deep prefetch into depth 1.  Protocol for users:
//   * every load issued with global_load16_async is consumed only after wait_vmcnt<N>() (N = loads issued after it that may still
//     be in flight: loads return in order) followed by settle() on its register;
//   * every such load IS consumed that way (or drained with wait_vmcnt<0>() + settle()): a destination register whose value is
//     never used is free for the allocator while the load is still in flight.
// Build switch (-DS2M2_UNTRACKED_LOADS=0): every default use becomes a TRACKED load and the counted waits no-ops -- slower (the
// compiler sinks the requests towards their uses) but independent of the scheduling assumptions tools/check_isa.py guards; what
// __graft_entry__.build() falls back to under S2M2_ISA_FALLBACK=1 when a compiler fails that check.
#ifndef S2M2_UNTRACKED_LOADS
#define S2M2_UNTRACKED_LOADS 1
#endif
template <bool ASYNC = (S2M2_UNTRACKED_LOADS != 0)>
__device__ __forceinline__ void global_load16_async(raw16_t& dst, const void* p) {
    if constexpr (ASYNC) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p));   // no "memory" clobber: it would pin every LDS access around it
    else dst = global_load16(p);                                  // tracked: the counted waits become no-ops
}
template <int N, bool ASYNC = (S2M2_UNTRACKED_LOADS != 0)> __device__ __forceinline__ void wait_vmcnt() {
    if constexpr (ASYNC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N));   // ordered against the loads / settle() by volatility alone
}
__device__ __forceinline__ void settle(raw16_t& v) { asm volatile("" : "+v"(v)); }
// counted wait whose count is a constant only after unrolling (the tail of a fully unrolled ring: fewer and fewer requests are younger than
// the fragment consumed next); n is clamped to [0, 8]
__device__ __forceinline__ void wait_vmcnt_n(int n) {
    switch (n) {
        case 8: wait_vmcnt<8>(); break;
        case 7: wait_vmcnt<7>(); break;
        case 6: wait_vmcnt<6>(); break;
        case 5: wait_vmcnt<5>(); break;
        case 4: wait_vmcnt<4>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 1: wait_vmcnt<1>(); break;
        default: wait_vmcnt<0>(); break;
    }
}

// flat thread index -> (gid / d, gid % d) in 32-bit arithmetic: a 64-bit division costs ~80 VALU instructions on gfx950, a 32-bit one ~25, and the
// one-thread-per-element kernels (K3, K7, K8) decode two or three of them per thread -- for the lookups that was more than the work itself.
// Callers guarantee gid < 2^31 (the host entry points check the element count).
__device__ __forceinline__ void divmod32(long long gid, int d, int& quot, int& rem) {
    const unsigned g = (unsigned)gid, q = g / (unsigned)d;
    quot = (int)q;
    rem = (int)(g - q * (unsigned)d);
}

// 16-byte vector of T (8 halfs / 4 floats)
template <typename T> struct Vec16;
template <> struct alignas(16) Vec16<half_t> { half_t v[8]; };
template <> struct alignas(16) Vec16<float>  { float v[4]; };

// ------------------------------------------------------------------------------------------------
// MFMA wrappers.  One "k16 fragment" = 8 consecutive k elements of one row (lanes 0-31: k 0..7 of row lane,
// lanes 32-63: k 8..15 of row lane-32) for BOTH dtypes: fp16 feeds one v_mfma_f32_32x32x16_f16, fp32 feeds
// eight v_mfma_f32_32x32x2_f32 (element s of both operands in step s; the sum over k is order independent,
// so relabelling k = 8*half + s -> step s, half is legal as long as A and B use the same map).
// Accumulator layout (both): lane l holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31], r = 0..15.
// ------------------------------------------------------------------------------------------------
template <typename T> struct Frag;
template <> struct Frag<half_t> { half8_t v; };
template <> struct Frag<float>  { float v[8]; };

__device__ __forceinline__ void mma32(float16_t& acc, const Frag<half_t>& a, const Frag<half_t>& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.v, b.v, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(float16_t& acc, const Frag<float>& a, const Frag<float>& b) {
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s], b.v[s], acc, 0, 0, 0);
}

// fragment <- LDS/global memory holding 8 consecutive T (16 B for half, 32 B for float), 16-B aligned
__device__ __forceinline__ void load_frag(Frag<half_t>& f, const half_t* p) { f.v = *reinterpret_cast<const half8_t*>(p); }
__device__ __forceinline__ void load_frag(Frag<float>& f, const float* p) {
    float4_t a = *reinterpret_cast<const float4_t*>(p);
    float4_t b = *reinterpret_cast<const float4_t*>(p + 4);
    f.v[0] = a[0]; f.v[1] = a[1]; f.v[2] = a[2]; f.v[3] = a[3];
    f.v[4] = b[0]; f.v[5] = b[1]; f.v[6] = b[2]; f.v[7] = b[3];
}

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------------------------------------
// wave-level helpers (wave = 64 lanes)
// ------------------------------------------------------------------------------------------------
template <int WIDTH> __device__ __forceinline__ float group_sum(float x) {
#pragma unroll
    for (int o = WIDTH / 2; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
}
// DPP cross-lane moves (VALU data path, no LDS round trip -- __shfl_xor lowers to ds_bpermute, ~100+ cycles each)
template <int CTRL> __device__ __forceinline__ float dpp_mov(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// sum over aligned groups of 8 lanes, result in all 8: quad_perm[1,0,3,2], quad_perm[2,3,0,1], row_half_mirror
__device__ __forceinline__ float group8_sum(float x) {
    x += dpp_mov<0xB1>(x);
    x += dpp_mov<0x4E>(x);
    x += dpp_mov<0x141>(x);
    return x;
}
__device__ __forceinline__ float wave_sum(float x) { return group_sum<64>(x); }
__device__ __forceinline__ float wave_max(float x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x = fmaxf(x, __shfl_xor(x, o, 64));
    return x;
}

// XCD-aware, bijective remap of a linear block id so that consecutive logical ids share an XCD (and its L2):
// hardware places block b on XCD b % 8 (observed, speed only).
__device__ __forceinline__ int xcd_remap(int bid, int nblocks) {
    const int q = nblocks >> 3, r = nblocks & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

}  // namespace s2m2
