// K3 -- two-level, radius-r cost-volume lookup for the refinement loop (one thread per pixel and tap).
//
// Replaces CostVolume.__init__ (avg_pool2d copy of the volume) and CostVolume.__call__ (two F.grid_sample calls)
// (/root/reference/src/s2m2/core/model/submodules.py:7-60; SURVEY.md A9+A10, Appendix A step 11).
// The half-resolution level is averaged on the fly from cv (no +50 % copy).  The reference goes pixel ->
// normalised (Python, `2*x/(W-1)-1`) -> pixel (ATen CPU kernel, `(g+1)*((size-1)/2)`) in fp32 on BOTH axes,
// so the effective coordinate is off by ~1 ulp and a vanishing weight can land on the neighbouring row;
// that arithmetic is reproduced here with contraction disabled.
#include "common.h"
#include "plan.h"
#include "cv_lookup.h"

namespace s2m2 {

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cv_lookup_kernel(const TI* __restrict__ cv, const float* __restrict__ disp,
                                                        TO* __restrict__ corr1, TO* __restrict__ corr2, int B, int h, int w,
                                                        int radius, long long batch_stride, long long pix_stride,
                                                        long long tap_stride, int pitch) {
    // grid: x = (pixel column, tap) of one image row, y = image row (b * h + y): 32-bit index arithmetic only (round 4: the flat 64-bit
    // index of rounds 1-3 cost five 64-bit divisions per thread -- more than the lookup itself)
    const int T = 2 * radius + 1;
    const unsigned t2 = blockIdx.x * blockDim.x + threadIdx.x;       // i * 2T + tap
    if (t2 >= (unsigned)(w * 2 * T)) return;
    const int i = (int)(t2 / (unsigned)(2 * T));
    const int t = (int)(t2 - (unsigned)i * (unsigned)(2 * T));
    const int rowid = blockIdx.y;                         // b*h + y
    const int b = rowid / h;
    const long long pix = (long long)rowid * w + i;       // (b*h + y)*w + i
    const int level = t >= T;
    const int k = level ? t - T : t;
    const TI* img = cv + (size_t)rowid * w * pitch;       // (w rows = left pixel i, `pitch` elements apart) x (w cols = right pixel j)
    const float out = lookup_tap<TI>(img, i, disp[pix], level, k, radius, w, pitch);
    TO* dst = (level ? corr2 : corr1) + (size_t)b * batch_stride + (size_t)(pix - (long long)b * h * w) * pix_stride + (size_t)k * tap_stride;
    *dst = from_f32<TO>(out);
}

template <typename TI, typename TO>
static int launch_lookup(const void* cv, const float* disp, void* c1, void* c2, int B, int h, int w, int radius, long long bs,
                         long long ps, long long ts, int pitch, hipStream_t st) {
    const int per_row = w * 2 * (2 * radius + 1);
    hipLaunchKernelGGL((cv_lookup_kernel<TI, TO>), dim3((per_row + 255) / 256, B * h), dim3(256), 0, st, static_cast<const TI*>(cv), disp,
                       static_cast<TO*>(c1), static_cast<TO*>(c2), B, h, w, radius, bs, ps, ts, pitch);
    return check_launch("cv_lookup");
}

}  // namespace s2m2

static int cv_lookup_impl(const void* cv, const float* disp, void* corr1, void* corr2, int B, int h, int w, int radius,
                              int cv_dtype, int out_dtype, long long batch_stride, long long pix_stride, long long tap_stride,
                              int cv_pitch, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(cv && disp && corr1 && corr2, "cv_lookup: null pointer");
    S2M2_REQUIRE(B > 0 && h > 0 && w > 1 && radius >= 0 && radius <= 16 && (long long)B * h <= 65535, "cv_lookup: bad arguments (B * h <= 65535 image rows per launch)");
    S2M2_REQUIRE(w % 2 == 0, "cv_lookup: w=%d must be even", w);
    if (cv_pitch == 0) cv_pitch = w;
    S2M2_REQUIRE(cv_pitch >= w, "cv_lookup: cv_pitch=%d must be at least w=%d", cv_pitch, w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cv_dtype == S2M2_F16 && out_dtype == S2M2_F16) return launch_lookup<half_t, half_t>(cv, disp, corr1, corr2, B, h, w, radius, batch_stride, pix_stride, tap_stride, cv_pitch, st);
    if (cv_dtype == S2M2_F16 && out_dtype == S2M2_F32) return launch_lookup<half_t, float>(cv, disp, corr1, corr2, B, h, w, radius, batch_stride, pix_stride, tap_stride, cv_pitch, st);
    if (cv_dtype == S2M2_F32 && out_dtype == S2M2_F32) return launch_lookup<float, float>(cv, disp, corr1, corr2, B, h, w, radius, batch_stride, pix_stride, tap_stride, cv_pitch, st);
    if (cv_dtype == S2M2_F32 && out_dtype == S2M2_F16) return launch_lookup<float, half_t>(cv, disp, corr1, corr2, B, h, w, radius, batch_stride, pix_stride, tap_stride, cv_pitch, st);
    return set_error("cv_lookup: unsupported dtypes cv=%d out=%d", cv_dtype, out_dtype);
}
extern "C" int s2m2_cv_lookup(const void* cv, const float* disp, void* corr1, void* corr2, int B, int h, int w, int radius,
                              int cv_dtype, int out_dtype, long long batch_stride, long long pix_stride, long long tap_stride,
                              int cv_pitch, void* stream) {
    return s2m2::plan_dispatch("s2m2_cv_lookup", &cv_lookup_impl, stream, cv, disp, corr1, corr2, B, h, w, radius, cv_dtype, out_dtype, batch_stride, pix_stride, tap_stride, cv_pitch);
}

