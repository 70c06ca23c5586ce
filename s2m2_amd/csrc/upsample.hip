// K7 -- convex upsampling: 9-way softmax of the mask logits + weighted sum of the 3x3 replicate-padded neighbourhood.
//
// Replaces S2M2.upsample4x / upsample1x (reference s2m2.py:101-133: custom_unfold (utils.py:9-20) -> F.interpolate(nearest)
// -> softmax(dim=1) -> multiply -> sum; with output_upsample also a bilinear x2 of the logits) for all three maps
// (disparity, occlusion, confidence) in one pass: the reference materialises three (B,9,H,W) unfolded tensors and a
// (B,9,H,W) softmax; here each output pixel reads its 9 logits once (NHWC, channels 0..8 of a 16-channel-padded row) and
// gathers the 9 low-resolution neighbours from cache.  HBM-bound: ~(logit row + 3*4 B) per output pixel.
//   out[m][b,Y,X] = scale_m * sum_n softmax_n(logit[b,Y,X,:9]) * x[m][b, clamp(Y/f + n/3 - 1), clamp(X/f + n%3 - 1)]
#include "common.h"
#include "plan.h"

namespace s2m2 {

struct UpArgs {
    const float* x[3];
    float* out[3];
    float scale[3];
    const void* logits;
    int nmaps, logit_stride, B, hs, ws, factor, logit_up2;
    void* chan_out;            // optional: map 0 also stored in the activation dtype at chan_out[gid * chan_stride]
    long long chan_stride;
};

template <typename T>
__device__ __forceinline__ void load9(const T* p, float (&l)[9]) {
    constexpr int VEC = 16 / sizeof(T);
    constexpr int NP = (9 + VEC - 1) / VEC;
    Vec16<T> v[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) v[q] = *reinterpret_cast<const Vec16<T>*>(p + q * VEC);
#pragma unroll
    for (int n = 0; n < 9; ++n) l[n] = to_f32(v[n / VEC].v[n % VEC]);
}

template <typename T>
__global__ __launch_bounds__(256) void convex_upsample_kernel(UpArgs a) {
    const int Ho = a.hs * a.factor, Wo = a.ws * a.factor;
    const long long total = (long long)a.B * Ho * Wo;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    int X, Y, b, t;
    divmod32(gid, Wo, t, X);
    divmod32(t, Ho, b, Y);
    const T* lg = static_cast<const T*>(a.logits);
    float l[9];
    if (!a.logit_up2) {
        load9<T>(lg + gid * a.logit_stride, l);
    } else {
        // bilinear x2, align_corners=False (ATen upsample_bilinear2d: src = max((dst + 0.5) * 0.5 - 0.5, 0)), rounded to T
        const float sy = fmaxf(((float)Y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf(((float)X + 0.5f) * 0.5f - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < a.hs - 1), x1 = x0 + (x0 < a.ws - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        float l00[9], l01[9], l10[9], l11[9];
        const long long base = (long long)b * a.hs * a.ws;
        load9<T>(lg + (base + (long long)y0 * a.ws + x0) * a.logit_stride, l00);
        load9<T>(lg + (base + (long long)y0 * a.ws + x1) * a.logit_stride, l01);
        load9<T>(lg + (base + (long long)y1 * a.ws + x0) * a.logit_stride, l10);
        load9<T>(lg + (base + (long long)y1 * a.ws + x1) * a.logit_stride, l11);
#pragma unroll
        for (int n = 0; n < 9; ++n) {
            const float v = (1.f - ly) * ((1.f - lx) * l00[n] + lx * l01[n]) + ly * ((1.f - lx) * l10[n] + lx * l11[n]);
            l[n] = to_f32(from_f32<T>(v));
        }
    }
    float mx = l[0];
#pragma unroll
    for (int n = 1; n < 9; ++n) mx = fmaxf(mx, l[n]);
    float den = 0.f;
#pragma unroll
    for (int n = 0; n < 9; ++n) { l[n] = expf(l[n] - mx); den += l[n]; }
    const float inv = 1.0f / den;
    const int yc = Y / a.factor, xc = X / a.factor;
    int yy[3], xx[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        int v = yc + d - 1; yy[d] = v < 0 ? 0 : (v > a.hs - 1 ? a.hs - 1 : v);
        v = xc + d - 1;     xx[d] = v < 0 ? 0 : (v > a.ws - 1 ? a.ws - 1 : v);
    }
    for (int m = 0; m < a.nmaps; ++m) {
        const float* xm = a.x[m] + (long long)b * a.hs * a.ws;
        float acc = 0.f;
#pragma unroll
        for (int n = 0; n < 9; ++n) acc += xm[(long long)yy[n / 3] * a.ws + xx[n % 3]] * (l[n] * inv);
        a.out[m][gid] = acc * a.scale[m];
        if (m == 0 && a.chan_out) static_cast<T*>(a.chan_out)[gid * a.chan_stride] = from_f32<T>(acc * a.scale[0]);
    }
}


// ---------------------------------------------------------------------------------------------------------------
// 2x resampling of NHWC activations: AvgPool2d(2) (unet.py:25-30, stacked_MRT.py:22-27) and bilinear x2,
// align_corners=False (unet.py:32-37, stacked_MRT.py:29-34; ATen upsample_bilinear2d arithmetic in fp32).
// One 16-byte channel piece per thread; HBM-bound.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int MODE>
__global__ __launch_bounds__(256) void resample2x_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C,
                                                         long long xs, long long ys) {
    constexpr int VEC = 16 / sizeof(T);
    const int P = C / VEC;
    const int Ho = MODE == 0 ? H / 2 : H * 2, Wo = MODE == 0 ? W / 2 : W * 2;
    const long long total = (long long)N * Ho * Wo * P;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    int pc, X, Y, n, t, t1;
    divmod32(gid, P, t, pc);
    divmod32(t, Wo, t1, X);
    divmod32(t1, Ho, n, Y);
    const T* xb = x + (long long)n * H * W * xs + pc * VEC;
    Vec16<T> o;
    if (MODE == 0) {
        const Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(xb + ((long long)(2 * Y) * W + 2 * X) * xs);
        const Vec16<T> b = *reinterpret_cast<const Vec16<T>*>(xb + ((long long)(2 * Y) * W + 2 * X + 1) * xs);
        const Vec16<T> c = *reinterpret_cast<const Vec16<T>*>(xb + ((long long)(2 * Y + 1) * W + 2 * X) * xs);
        const Vec16<T> d = *reinterpret_cast<const Vec16<T>*>(xb + ((long long)(2 * Y + 1) * W + 2 * X + 1) * xs);
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>((to_f32(a.v[e]) + to_f32(b.v[e]) + to_f32(c.v[e]) + to_f32(d.v[e])) * 0.25f);
    } else {
        const float sy = fmaxf(((float)Y + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf(((float)X + 0.5f) * 0.5f - 0.5f, 0.f);
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        const Vec16<T> a = *reinterpret_cast<const Vec16<T>*>(xb + ((long long)y0 * W + x0) * xs);
        const Vec16<T> b = *reinterpret_cast<const Vec16<T>*>(xb + ((long long)y0 * W + x1) * xs);
        const Vec16<T> c = *reinterpret_cast<const Vec16<T>*>(xb + ((long long)y1 * W + x0) * xs);
        const Vec16<T> d = *reinterpret_cast<const Vec16<T>*>(xb + ((long long)y1 * W + x1) * xs);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            o.v[e] = from_f32<T>(hy * (hx * to_f32(a.v[e]) + lx * to_f32(b.v[e])) + ly * (hx * to_f32(c.v[e]) + lx * to_f32(d.v[e])));
    }
    *reinterpret_cast<Vec16<T>*>(y + (((long long)n * Ho + Y) * Wo + X) * ys + pc * VEC) = o;
}

}  // namespace s2m2

static int convex_upsample_impl(const float* const* x, float* const* out, const float* scale, int nmaps, const void* logits,
                                    int logit_stride, int B, int hs, int ws, int factor, int logit_up2, void* chan_out,
                                    long long chan_stride, int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(x && out && logits && scale, "convex_upsample: null pointer");
    S2M2_REQUIRE(nmaps >= 1 && nmaps <= 3, "convex_upsample: nmaps=%d (1..3)", nmaps);
    S2M2_REQUIRE(B > 0 && hs > 0 && ws > 0 && factor >= 1, "convex_upsample: bad shape");
    S2M2_REQUIRE((long long)B * hs * factor * ws * factor < (1LL << 31), "convex_upsample: more than 2^31 output pixels");
    S2M2_REQUIRE(logit_stride >= 16 && logit_stride % 8 == 0, "convex_upsample: logit rows must be padded to >= 16 channels (stride %d)", logit_stride);
    S2M2_REQUIRE(!logit_up2 || factor == 2, "convex_upsample: logit_up2 needs factor 2");
    UpArgs a;
    for (int m = 0; m < 3; ++m) {
        a.x[m] = m < nmaps ? x[m] : nullptr;
        a.out[m] = m < nmaps ? out[m] : nullptr;
        a.scale[m] = m < nmaps ? scale[m] : 0.f;
        if (m < nmaps) S2M2_REQUIRE(x[m] && out[m], "convex_upsample: map %d is null", m);
    }
    a.logits = logits; a.nmaps = nmaps; a.logit_stride = logit_stride; a.B = B; a.hs = hs; a.ws = ws; a.factor = factor;
    a.logit_up2 = logit_up2;
    a.chan_out = chan_out; a.chan_stride = chan_stride;
    const long long total = (long long)B * hs * factor * ws * factor;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == S2M2_F16) hipLaunchKernelGGL((convex_upsample_kernel<half_t>), grid, dim3(256), 0, st, a);
    else if (dtype == S2M2_F32) hipLaunchKernelGGL((convex_upsample_kernel<float>), grid, dim3(256), 0, st, a);
    else return set_error("convex_upsample: unsupported dtype %d", dtype);
    return check_launch("convex_upsample");
}
// Plans record arguments BY VALUE: the three host arrays of this entry point (device pointers of the maps, scales) are copied into a
// descriptor whose words the plan scans and patches; the trampoline rebuilds the arrays from it.
struct UpsampleBlob {
    const float* x[3]; float* out[3]; float scale[3];
    int nmaps; const void* logits; int logit_stride, B, hs, ws, factor, logit_up2; void* chan_out; long long chan_stride; int dtype;
};
namespace s2m2 {
S2M2_PLAN_PTRS(UpsampleBlob, S2M2_OFF_I(UpsampleBlob, x, 0), S2M2_OFF_I(UpsampleBlob, x, 1), S2M2_OFF_I(UpsampleBlob, x, 2), S2M2_OFF_I(UpsampleBlob, out, 0),
               S2M2_OFF_I(UpsampleBlob, out, 1), S2M2_OFF_I(UpsampleBlob, out, 2), S2M2_OFF(UpsampleBlob, logits), S2M2_OFF(UpsampleBlob, chan_out))
}
static int convex_upsample_blob(const UpsampleBlob* b, void* stream) {
    return convex_upsample_impl(b->x, b->out, b->scale, b->nmaps, b->logits, b->logit_stride, b->B, b->hs, b->ws, b->factor, b->logit_up2,
                                b->chan_out, b->chan_stride, b->dtype, stream);
}
extern "C" int s2m2_convex_upsample(const float* const* x, float* const* out, const float* scale, int nmaps, const void* logits,
                                    int logit_stride, int B, int hs, int ws, int factor, int logit_up2, void* chan_out,
                                    long long chan_stride, int dtype, void* stream) {
    if (!x || !out || !scale || nmaps < 1 || nmaps > 3)            // (the impl reports these; nothing to copy by value)
        return convex_upsample_impl(x, out, scale, nmaps, logits, logit_stride, B, hs, ws, factor, logit_up2, chan_out, chan_stride, dtype, stream);
    UpsampleBlob b;
    __builtin_memset(&b, 0, sizeof(b));
    for (int m = 0; m < nmaps; ++m) { b.x[m] = x[m]; b.out[m] = out[m]; b.scale[m] = scale[m]; }
    b.nmaps = nmaps; b.logits = logits; b.logit_stride = logit_stride; b.B = B; b.hs = hs; b.ws = ws; b.factor = factor;
    b.logit_up2 = logit_up2; b.chan_out = chan_out; b.chan_stride = chan_stride; b.dtype = dtype;
    return s2m2::plan_dispatch_desc("s2m2_convex_upsample", &convex_upsample_blob, &b, stream);
}


static int resample2x_impl(const void* x, void* y, int N, int H, int W, int C, long long x_stride, long long y_stride, int mode,
                               int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(x && y, "resample2x: null pointer");
    S2M2_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && x_stride % 8 == 0 && y_stride % 8 == 0, "resample2x: bad shape");
    S2M2_REQUIRE(mode == 0 || mode == 1, "resample2x: mode %d (0 = average pool 2x2, 1 = bilinear x2)", mode);
    S2M2_REQUIRE((long long)N * H * W * 4 * (C / 4) < (1LL << 31), "resample2x: more than 2^31 output pieces");
    S2M2_REQUIRE(mode == 1 || (H % 2 == 0 && W % 2 == 0), "resample2x: average pooling needs even H, W");
    const int vec = dtype == S2M2_F16 ? 8 : 4;
    const long long Ho = mode == 0 ? H / 2 : H * 2, Wo = mode == 0 ? W / 2 : W * 2;
    const long long total = (long long)N * Ho * Wo * (C / vec);
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == S2M2_F16 && mode == 0) hipLaunchKernelGGL((resample2x_kernel<half_t, 0>), grid, dim3(256), 0, st, (const half_t*)x, (half_t*)y, N, H, W, C, x_stride, y_stride);
    else if (dtype == S2M2_F16) hipLaunchKernelGGL((resample2x_kernel<half_t, 1>), grid, dim3(256), 0, st, (const half_t*)x, (half_t*)y, N, H, W, C, x_stride, y_stride);
    else if (dtype == S2M2_F32 && mode == 0) hipLaunchKernelGGL((resample2x_kernel<float, 0>), grid, dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, x_stride, y_stride);
    else if (dtype == S2M2_F32) hipLaunchKernelGGL((resample2x_kernel<float, 1>), grid, dim3(256), 0, st, (const float*)x, (float*)y, N, H, W, C, x_stride, y_stride);
    else return set_error("resample2x: unsupported dtype %d", dtype);
    return check_launch("resample2x");
}
extern "C" int s2m2_resample2x(const void* x, void* y, int N, int H, int W, int C, long long x_stride, long long y_stride, int mode,
                               int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_resample2x", &resample2x_impl, stream, x, y, N, H, W, C, x_stride, y_stride, mode, dtype);
}

