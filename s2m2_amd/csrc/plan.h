// Recorded launch plans (s2m2_plan_*, s2m2_refine_step): see include/s2m2_hip.h.  Every launch-type entry point of the library goes through
// plan_dispatch / plan_dispatch_desc: the call is made as always, and -- while the calling thread records -- its arguments are appended to the
// plan as a flat blob (positional arguments as a trivially copyable pack, descriptors by value) together with a trampoline that re-issues it.
// Every blob carries a mask of its POINTER words (ABI 600, round 6): s2m2_plan_end looks for pointers into the external buffers in those words
// only -- an int pair, a stride or a size that happens to fall inside an external's address range is never rewritten.  Positional packs derive
// the mask from the argument types; descriptors list their pointer fields below (PlanPtrFields: a descriptor without a list does not compile).
#pragma once
#include <stddef.h>
#include <type_traits>
#include "common.h"

namespace s2m2 {

bool plan_recording();
constexpr int kPlanMaskWords = 4;                                // 256 eight-byte words: blobs of up to 2 KB
struct PlanPtrMask {
    unsigned long long bits[kPlanMaskWords] = {0, 0, 0, 0};
    bool overflow = false;
    void set(size_t byte_off) {
        const size_t w = byte_off / 8;
        if (byte_off % 8 != 0 || w >= 64 * kPlanMaskWords) overflow = true;
        else bits[w / 64] |= 1ULL << (w % 64);
    }
    bool test(size_t w) const { return w < 64 * kPlanMaskWords && ((bits[w / 64] >> (w % 64)) & 1ULL) != 0; }
};
int plan_append(int (*tramp)(const void* blob, void* stream), const void* blob, size_t bytes, const char* name, const PlanPtrMask& mask);

// positional arguments as a plain aggregate (std::tuple is not guaranteed trivially copyable; the blobs are copied and scanned as raw words)
template <typename... A> struct ArgPack;
template <> struct ArgPack<> {};
template <typename H, typename... T> struct ArgPack<H, T...> { H head; ArgPack<T...> tail; };

template <typename... A> struct PlanBlob { int (*impl)(A..., void*); ArgPack<A...> args; };

template <typename F, typename... Done>
inline int apply_pack(F&& f, const ArgPack<>&, Done... done) { return f(done...); }
template <typename F, typename H, typename... T, typename... Done>
inline int apply_pack(F&& f, const ArgPack<H, T...>& p, Done... done) { return apply_pack(f, p.tail, done..., p.head); }

template <typename... A> int plan_tramp(const void* blob, void* stream) {
    const auto* b = static_cast<const PlanBlob<A...>*>(blob);
    return apply_pack([&](A... x) { return b->impl(x..., stream); }, b->args);
}
template <typename... A> inline void fill_pack(ArgPack<A...>&) {}
template <typename H, typename... T> inline void fill_pack(ArgPack<H, T...>& p, H h, T... t) { p.head = h; fill_pack(p.tail, t...); }
// the pointer-typed arguments of a pack -> mask bits (offsets relative to the blob that holds the pack)
inline void mark_pack(const void*, const ArgPack<>&, PlanPtrMask&) {}
template <typename H, typename... T> inline void mark_pack(const void* blob, const ArgPack<H, T...>& p, PlanPtrMask& m) {
    if (std::is_pointer<H>::value) m.set((size_t)((const char*)&p.head - (const char*)blob));
    mark_pack(blob, p.tail, m);
}

template <typename... A> int plan_dispatch(const char* name, int (*impl)(A..., void*), void* stream, A... a) {
#if S2M2_RANGE_CHECK
    range_bind_tu();
    const int rc = impl(a..., stream) || range_collect(name, stream);
#else
    const int rc = impl(a..., stream);
#endif
    if (rc == 0 && plan_recording()) {
        PlanBlob<A...> b;
        __builtin_memset(&b, 0, sizeof(b));                        // padding words are scanned too: keep them deterministic
        b.impl = impl;
        fill_pack(b.args, a...);
        PlanPtrMask m;
        mark_pack(&b, b.args, m);
        return plan_append(&plan_tramp<A...>, &b, sizeof(b), name, m);
    }
    return rc;
}

// pointer fields of the descriptors that go through plans: byte offsets inside the descriptor
template <typename D> struct PlanPtrFields;                        // (no primary definition: every recorded descriptor type lists its pointers)
#define S2M2_PLAN_PTRS(D, ...)                                                                                      \
    template <> struct PlanPtrFields<D> {                                                                           \
        static void mark(size_t base, PlanPtrMask& m) {                                                             \
            const size_t offs[] = {__VA_ARGS__};                                                                    \
            for (size_t o : offs) m.set(base + o);                                                                  \
        }                                                                                                           \
    };
#define S2M2_OFF(D, f) offsetof(D, f)
#define S2M2_OFF_I(D, f, i) (offsetof(D, f) + 8 * (i))
S2M2_PLAN_PTRS(s2m2_corr_desc, S2M2_OFF(s2m2_corr_desc, tokens), S2M2_OFF(s2m2_corr_desc, ln_weight), S2M2_OFF(s2m2_corr_desc, ln_bias),
               S2M2_OFF(s2m2_corr_desc, cv), S2M2_OFF(s2m2_corr_desc, start_event), S2M2_OFF(s2m2_corr_desc, stop_event))
S2M2_PLAN_PTRS(s2m2_conv_desc, S2M2_OFF_I(s2m2_conv_desc, src, 0), S2M2_OFF_I(s2m2_conv_desc, src, 1), S2M2_OFF_I(s2m2_conv_desc, src, 2),
               S2M2_OFF_I(s2m2_conv_desc, src, 3), S2M2_OFF(s2m2_conv_desc, weight), S2M2_OFF(s2m2_conv_desc, bias), S2M2_OFF(s2m2_conv_desc, out),
               S2M2_OFF(s2m2_conv_desc, aux0), S2M2_OFF(s2m2_conv_desc, aux1), S2M2_OFF(s2m2_conv_desc, ln_wsum), S2M2_OFF(s2m2_conv_desc, bias2))
S2M2_PLAN_PTRS(s2m2_chain_desc, S2M2_OFF(s2m2_chain_desc, x), S2M2_OFF(s2m2_chain_desc, res), S2M2_OFF(s2m2_chain_desc, out),
               S2M2_OFF_I(s2m2_chain_desc, weight, 0), S2M2_OFF_I(s2m2_chain_desc, weight, 1), S2M2_OFF_I(s2m2_chain_desc, weight, 2),
               S2M2_OFF_I(s2m2_chain_desc, bias, 0), S2M2_OFF_I(s2m2_chain_desc, bias, 1), S2M2_OFF_I(s2m2_chain_desc, bias, 2),
               S2M2_OFF_I(s2m2_chain_desc, ln_wsum, 0), S2M2_OFF_I(s2m2_chain_desc, ln_wsum, 1), S2M2_OFF_I(s2m2_chain_desc, ln_wsum, 2),
               S2M2_OFF(s2m2_chain_desc, ln_out), S2M2_OFF(s2m2_chain_desc, ln_gamma), S2M2_OFF(s2m2_chain_desc, ln_beta),
               S2M2_OFF(s2m2_chain_desc, fan_weight), S2M2_OFF(s2m2_chain_desc, fan_bias), S2M2_OFF(s2m2_chain_desc, fan_ln_wsum),
               S2M2_OFF(s2m2_chain_desc, fan_out))
S2M2_PLAN_PTRS(s2m2_rowattn_desc, S2M2_OFF(s2m2_rowattn_desc, x), S2M2_OFF(s2m2_rowattn_desc, out), S2M2_OFF(s2m2_rowattn_desc, weights),
               S2M2_OFF(s2m2_rowattn_desc, vectors), S2M2_OFF(s2m2_rowattn_desc, ln_out))
S2M2_PLAN_PTRS(s2m2_convblock_desc, S2M2_OFF(s2m2_convblock_desc, x), S2M2_OFF(s2m2_convblock_desc, out), S2M2_OFF(s2m2_convblock_desc, w_conv0),
               S2M2_OFF(s2m2_convblock_desc, w_conv2), S2M2_OFF(s2m2_convblock_desc, w_1x0), S2M2_OFF(s2m2_convblock_desc, w_1x2),
               S2M2_OFF(s2m2_convblock_desc, b_conv0), S2M2_OFF(s2m2_convblock_desc, b_conv2), S2M2_OFF(s2m2_convblock_desc, b_1x0),
               S2M2_OFF(s2m2_convblock_desc, b_1x2))
S2M2_PLAN_PTRS(s2m2_pw_desc, S2M2_OFF_I(s2m2_pw_desc, src, 0), S2M2_OFF_I(s2m2_pw_desc, src, 1), S2M2_OFF_I(s2m2_pw_desc, src, 2),
               S2M2_OFF_I(s2m2_pw_desc, src, 3), S2M2_OFF(s2m2_pw_desc, weight_frag), S2M2_OFF(s2m2_pw_desc, bias), S2M2_OFF(s2m2_pw_desc, out))
S2M2_PLAN_PTRS(s2m2_narrow_desc, S2M2_OFF(s2m2_narrow_desc, x), S2M2_OFF(s2m2_narrow_desc, x1), S2M2_OFF(s2m2_narrow_desc, weight_frag),
               S2M2_OFF(s2m2_narrow_desc, bias), S2M2_OFF(s2m2_narrow_desc, out), S2M2_OFF(s2m2_narrow_desc, head_frag),
               S2M2_OFF(s2m2_narrow_desc, head_bias))

template <typename D> struct PlanDescBlob { int (*impl)(const D*, void*); D desc; };
template <typename D> int plan_desc_tramp(const void* blob, void* stream) {
    const auto* b = static_cast<const PlanDescBlob<D>*>(blob);
    return b->impl(&b->desc, stream);
}
template <typename D> int plan_dispatch_desc(const char* name, int (*impl)(const D*, void*), const D* d, void* stream) {
#if S2M2_RANGE_CHECK
    range_bind_tu();
    const int rc = impl(d, stream) || range_collect(name, stream);
#else
    const int rc = impl(d, stream);
#endif
    if (rc == 0 && d && plan_recording()) {
        PlanDescBlob<D> b;
        __builtin_memset(&b, 0, sizeof(b));
        b.impl = impl;
        b.desc = *d;
        PlanPtrMask m;
        PlanPtrFields<D>::mark(offsetof(PlanDescBlob<D>, desc), m);
        return plan_append(&plan_desc_tramp<D>, &b, sizeof(b), name, m);
    }
    return rc;
}

}  // namespace s2m2
