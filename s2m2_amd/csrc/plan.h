// Recorded launch plans (s2m2_plan_*, s2m2_refine_step): see include/s2m2_hip.h.  Every launch-type entry point of the library goes through
// plan_dispatch / plan_dispatch_desc: the call is made as always, and -- while the calling thread records -- its arguments are appended to the
// plan as a flat blob (positional arguments as a trivially copyable pack, descriptors by value) together with a trampoline that re-issues it.
#pragma once
#include <stddef.h>
#include "common.h"

namespace s2m2 {

bool plan_recording();
int plan_append(int (*tramp)(const void* blob, void* stream), const void* blob, size_t bytes, const char* name);

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
        return plan_append(&plan_tramp<A...>, &b, sizeof(b), name);
    }
    return rc;
}

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
        return plan_append(&plan_desc_tramp<D>, &b, sizeof(b), name);
    }
    return rc;
}

}  // namespace s2m2
