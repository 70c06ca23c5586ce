// K8 -- the small per-pixel stages between the big kernels of the forward, each fused into ONE launch (PyTorch runs every line
// of them as its own elementwise kernel: ~340 launches and ~1.2 ms per 1216x1024 pair, profiles/r01):
//   image_prep     normalize_img + left/right concat + NHWC/8-channel packing       (s2m2.py:80-89,140-143)
//   refine_prep    side inputs of GlobalRefiner / LocalRefiner                      (refinenet.py:63-68, 134-141)
//   global_update  disp = mask*disp + (1-mask)*update*100 [, clamp]                 (refinenet.py:70-71, s2m2.py:160-161)
//   refine_update  disp += d; conf/occ = sigmoid(d + logit); clamp; occ mask        (refinenet.py:149-151, s2m2.py:177-180)
//   unary          tanh of the context features (hidden state init)                 (s2m2.py:166)
// All maps are (B,h,w) fp32; "small" side inputs are (B,h,w,8) NHWC in the activation dtype with unused channels zero.
#include "common.h"
#include "plan.h"
#include "epilogue.h"

namespace s2m2 {

__device__ __forceinline__ float logit_eps(float p, float eps) {
    p = fminf(fmaxf(p, eps), 1.0f - eps);
    return logf(p / (1.0f - p));
}

template <typename TI, typename T>
__global__ __launch_bounds__(256) void image_prep_kernel(const TI* __restrict__ img0, const TI* __restrict__ img1, T* __restrict__ x8,
                                                         int B, long long HW) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= 2LL * B * HW) return;
    int ni, pixi;
    divmod32(gid, (int)HW, ni, pixi);
    const long long n = ni, pix = pixi;
    const TI* src = (n < B ? img0 + n * 3 * HW : img1 + (n - B) * 3 * HW) + pix;
    alignas(16) T o[8];
    o[0] = from_f32<T>(0.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) o[1 + c] = from_f32<T>(((float)src[c * HW] / 255.0f - 0.5f) * 2.0f);
#pragma unroll
    for (int c = 4; c < 8; ++c) o[c] = from_f32<T>(0.f);
    T* dst = x8 + gid * 8;
    if constexpr (sizeof(T) == 2) {
        *reinterpret_cast<Vec16<T>*>(dst) = *reinterpret_cast<Vec16<T>*>(o);
    } else {
        *reinterpret_cast<Vec16<T>*>(dst) = *reinterpret_cast<Vec16<T>*>(o);
        *reinterpret_cast<Vec16<T>*>(dst + 4) = *reinterpret_cast<Vec16<T>*>(o + 4);
    }
}

// mode 0 (global refiner): ch0 = disp/100*mask, ch1 = logit(mask*conf, 0.1), mask = conf > 0.2
// mode 1 (local refiner):  ch0 = disp/100, ch1 = logit(conf, 0.01), ch2 = logit(occ, 0.01)
template <typename T>
__global__ __launch_bounds__(256) void refine_prep_kernel(const float* __restrict__ disp, const float* __restrict__ conf,
                                                          const float* __restrict__ occ, T* __restrict__ small, long long n, int mode) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n) return;
    float v0, v1, v2 = 0.f;
    if (mode == 0) {
        const float c = conf[gid];
        const float mask = c > 0.2f ? 1.0f : 0.0f;
        v0 = disp[gid] / 1e2f * mask;
        v1 = logit_eps(mask * c, 1e-1f);
    } else {
        v0 = disp[gid] / 1e2f;
        v1 = logit_eps(conf[gid], 1e-2f);
        v2 = logit_eps(occ[gid], 1e-2f);
    }
    T* d = small + gid * 8;
    d[0] = from_f32<T>(v0); d[1] = from_f32<T>(v1); d[2] = from_f32<T>(v2);
#pragma unroll
    for (int c = 3; c < 8; ++c) d[c] = from_f32<T>(0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void global_update_kernel(const T* __restrict__ upd, int upd_stride, const float* __restrict__ disp,
                                                            const float* __restrict__ conf, float* __restrict__ out, long long n, int clamp0) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n) return;
    const float mask = conf[gid] > 0.2f ? 1.0f : 0.0f;
    float d = mask * disp[gid] + (1.0f - mask) * (to_f32(upd[gid * upd_stride]) * 1e2f);
    if (clamp0) d = fmaxf(d, 0.f);
    out[gid] = d;
}

// dco: (.., stride) with channel 0 = disparity delta, channels 8, 9 = confidence / occlusion logit deltas.  Out of place (the outputs
// may alias the inputs: every thread reads its own pixel before it writes it); small_next != nullptr: also the mode-1 side input
// of the NEXT refinement iteration (what refine_prep_kernel would compute from the values just written).
template <typename T>
__global__ __launch_bounds__(256) void refine_update_kernel(const T* __restrict__ dco, int stride, const float* disp, const float* conf,
                                                            const float* occ, float* disp_out, float* conf_out, float* occ_out,
                                                            T* __restrict__ small_next, long long n, int w, int use_pos) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= n) return;
    const T* r = dco + gid * stride;
    float d = disp[gid] + to_f32(r[0]);
    const float c = 1.0f / (1.0f + expf(-(to_f32(r[8]) + logit_eps(conf[gid], 1e-2f))));
    float o = 1.0f / (1.0f + expf(-(to_f32(r[9]) + logit_eps(occ[gid], 1e-2f))));
    if (use_pos) d = fmaxf(d, 0.f);
    const float x = (float)((unsigned)gid % (unsigned)w);
    o = (x - d >= 0.f) ? o : 0.f;
    disp_out[gid] = d; conf_out[gid] = c; occ_out[gid] = o;
    if (small_next) {
        alignas(16) T s8[8];
        s8[0] = from_f32<T>(d / 1e2f); s8[1] = from_f32<T>(logit_eps(c, 1e-2f)); s8[2] = from_f32<T>(logit_eps(o, 1e-2f));
#pragma unroll
        for (int k = 3; k < 8; ++k) s8[k] = from_f32<T>(0.f);
        T* dst = small_next + gid * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = s8[k];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void tanh_kernel(const T* __restrict__ x, T* __restrict__ y, long long npieces) {
    constexpr int VEC = 16 / sizeof(T);
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= npieces) return;
    const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(x + gid * VEC);
    Vec16<T> o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>(tanhf(to_f32(v.v[e])));
    *reinterpret_cast<Vec16<T>*>(y + gid * VEC) = o;
}

static inline dim3 grid1(long long n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace s2m2

static int image_prep_impl(const void* img0, const void* img1, void* x8, int B, int H, int W, int img_dtype, int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(img0 && img1 && x8 && B > 0 && H > 0 && W > 0 && 2LL * B * H * W < (1LL << 31), "image_prep: bad arguments (at most 2^31 pixels per launch)");
    const long long HW = (long long)H * W, n = 2LL * B * HW;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // img_dtype: S2M2_F32 / S2M2_F16 / 2 = uint8
    #define S2M2_IP(TI, T) hipLaunchKernelGGL((image_prep_kernel<TI, T>), grid1(n), dim3(256), 0, st, (const TI*)img0, (const TI*)img1, (T*)x8, B, HW)
    if (dtype == S2M2_F16) {
        if (img_dtype == S2M2_F32) S2M2_IP(float, half_t); else if (img_dtype == S2M2_F16) S2M2_IP(half_t, half_t);
        else if (img_dtype == 2) S2M2_IP(unsigned char, half_t); else return set_error("image_prep: unsupported image dtype %d", img_dtype);
    } else if (dtype == S2M2_F32) {
        if (img_dtype == S2M2_F32) S2M2_IP(float, float); else if (img_dtype == S2M2_F16) S2M2_IP(half_t, float);
        else if (img_dtype == 2) S2M2_IP(unsigned char, float); else return set_error("image_prep: unsupported image dtype %d", img_dtype);
    } else return set_error("image_prep: unsupported dtype %d", dtype);
    #undef S2M2_IP
    return check_launch("image_prep");
}
extern "C" int s2m2_image_prep(const void* img0, const void* img1, void* x8, int B, int H, int W, int img_dtype, int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_image_prep", &image_prep_impl, stream, img0, img1, x8, B, H, W, img_dtype, dtype);
}


static int refine_prep_impl(const float* disp, const float* conf, const float* occ, void* small8, long long npix, int mode, int dtype,
                                void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(disp && conf && small8 && npix > 0 && (mode == 0 || (mode == 1 && occ)), "refine_prep: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == S2M2_F16) hipLaunchKernelGGL((refine_prep_kernel<half_t>), grid1(npix), dim3(256), 0, st, disp, conf, occ, (half_t*)small8, npix, mode);
    else if (dtype == S2M2_F32) hipLaunchKernelGGL((refine_prep_kernel<float>), grid1(npix), dim3(256), 0, st, disp, conf, occ, (float*)small8, npix, mode);
    else return set_error("refine_prep: unsupported dtype %d", dtype);
    return check_launch("refine_prep");
}
extern "C" int s2m2_refine_prep(const float* disp, const float* conf, const float* occ, void* small8, long long npix, int mode, int dtype,
                                void* stream) {
    return s2m2::plan_dispatch("s2m2_refine_prep", &refine_prep_impl, stream, disp, conf, occ, small8, npix, mode, dtype);
}


static int global_update_impl(const void* upd, int upd_stride, const float* disp, const float* conf, float* out, long long npix,
                                  int clamp0, int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(upd && disp && conf && out && npix > 0 && upd_stride > 0, "global_update: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == S2M2_F16) hipLaunchKernelGGL((global_update_kernel<half_t>), grid1(npix), dim3(256), 0, st, (const half_t*)upd, upd_stride, disp, conf, out, npix, clamp0);
    else if (dtype == S2M2_F32) hipLaunchKernelGGL((global_update_kernel<float>), grid1(npix), dim3(256), 0, st, (const float*)upd, upd_stride, disp, conf, out, npix, clamp0);
    else return set_error("global_update: unsupported dtype %d", dtype);
    return check_launch("global_update");
}
extern "C" int s2m2_global_update(const void* upd, int upd_stride, const float* disp, const float* conf, float* out, long long npix,
                                  int clamp0, int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_global_update", &global_update_impl, stream, upd, upd_stride, disp, conf, out, npix, clamp0, dtype);
}


static int refine_update_to_impl(const void* dco, int dco_stride, const float* disp, const float* conf, const float* occ,
                                     float* disp_out, float* conf_out, float* occ_out, void* small8_next, long long npix, int w,
                                     int use_positivity, int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(dco && disp && conf && occ && disp_out && conf_out && occ_out && npix > 0 && w > 0 && dco_stride >= 10,
                 "refine_update: bad arguments");
    S2M2_REQUIRE(npix < (1LL << 31), "refine_update: 2^31 or more pixels (the column index is decoded in 32 bits)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == S2M2_F16)
        hipLaunchKernelGGL((refine_update_kernel<half_t>), grid1(npix), dim3(256), 0, st, (const half_t*)dco, dco_stride, disp, conf, occ,
                           disp_out, conf_out, occ_out, (half_t*)small8_next, npix, w, use_positivity);
    else if (dtype == S2M2_F32)
        hipLaunchKernelGGL((refine_update_kernel<float>), grid1(npix), dim3(256), 0, st, (const float*)dco, dco_stride, disp, conf, occ,
                           disp_out, conf_out, occ_out, (float*)small8_next, npix, w, use_positivity);
    else return set_error("refine_update: unsupported dtype %d", dtype);
    return check_launch("refine_update");
}
extern "C" int s2m2_refine_update_to(const void* dco, int dco_stride, const float* disp, const float* conf, const float* occ,
                                     float* disp_out, float* conf_out, float* occ_out, void* small8_next, long long npix, int w,
                                     int use_positivity, int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_refine_update_to", &refine_update_to_impl, stream, dco, dco_stride, disp, conf, occ, disp_out, conf_out, occ_out, small8_next, npix, w, use_positivity, dtype);
}


extern "C" int s2m2_refine_update(const void* dco, int dco_stride, float* disp, float* conf, float* occ, long long npix, int w,
                                  int use_positivity, int dtype, void* stream) {
    return s2m2_refine_update_to(dco, dco_stride, disp, conf, occ, disp, conf, occ, nullptr, npix, w, use_positivity, dtype, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// stem: the two 1x1 layers at the head of the CNN encoder, conv0 = Conv2d(3,16,1) - GELU - Conv2d(16,16,1) (reference
// submodules.py:68-71), on FULL-resolution pixels (2.5 M per pair at 1216x1024).  With K = 8 / 16 and N = 16 a GEMM tile is
// all padding and per-block overhead (K5: 145 + 134 us); per pixel it is 384 FMAs + 16 GELUs on the VALU: weights are wave-uniform
// (scalar loads), the 16-channel intermediate never leaves registers (rounded to the I/O dtype where the separate layers stored it),
// 16 bytes in, 32 bytes out per pixel.
// Round 6 (the pass was VALU-bound at 47.8 us, 2.5 TB/s): a thread takes TWO consecutive pixels as the halves of packed fp32 registers
// -- every multiply-add is a v_pk_fma_f32 with the weight broadcast from an SGPR, the GELU is the packed form as it stands -- and
// input columns whose 16 weights are all zero (5 of the 8: the tensor carries 3 image planes) are skipped behind a wave-uniform test
// (their terms are + 0 * x: bit-identical for finite x).  Same summation order, same rounding points as before.
// ---------------------------------------------------------------------------------------------------------------
namespace s2m2 {
template <typename T>
__global__ __launch_bounds__(256) void stem_mlp_kernel(const T* __restrict__ x8, const float* __restrict__ w0, const float* __restrict__ b0,
                                                       const float* __restrict__ w1, const float* __restrict__ b1, T* __restrict__ out,
                                                       long long npix) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= npix) return;
    const long long i1 = i + 1 < npix ? i + 1 : i;                 // (odd pixel count: the last thread computes its pixel twice)
    float2_t x[8];
    if constexpr (sizeof(T) == 2) {
        raw16_t ra = global_load16(x8 + i * 8), rb = global_load16(x8 + i1 * 8);           // (whole pieces: the columns' uses sit behind branches)
        asm volatile("" : "+v"(ra), "+v"(rb));
        const Vec16<T> va = __builtin_bit_cast(Vec16<T>, ra), vb = __builtin_bit_cast(Vec16<T>, rb);
#pragma unroll
        for (int k = 0; k < 8; ++k) x[k] = float2_t{to_f32(va.v[k]), to_f32(vb.v[k])};
    } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const Vec16<T> va = *reinterpret_cast<const Vec16<T>*>(x8 + i * 8 + 4 * q), vb = *reinterpret_cast<const Vec16<T>*>(x8 + i1 * 8 + 4 * q);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[4 * q + k] = float2_t{to_f32(va.v[k]), to_f32(vb.v[k])};
        }
    }
    float2_t h[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) h[j] = float2_t{b0[j], b0[j]};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        unsigned any = 0;                                          // (uniform integer arithmetic: scalar unit)
#pragma unroll
        for (int j = 0; j < 16; ++j) any |= __builtin_bit_cast(unsigned, w0[j * 8 + k]) << 1;
        if (any == 0) continue;                                    // a column of (+-) zeros
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float w = w0[j * 8 + k];
            h[j] = __builtin_elementwise_fma(float2_t{w, w}, x[k], h[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        if constexpr (sizeof(T) == 2 && S2M2_GELU16_POLY) h[j] = fast_gelu16x2(h[j]);
        else h[j] = float2_t{activate_to<S2M2_ACT_GELU, T>(h[j][0]), activate_to<S2M2_ACT_GELU, T>(h[j][1])};
        h[j] = float2_t{to_f32(from_f32<T>(h[j][0])), to_f32(from_f32<T>(h[j][1]))};
    }
    float2_t y[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        float2_t a = float2_t{b1[o], b1[o]};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float w = w1[o * 16 + j];
            a = __builtin_elementwise_fma(float2_t{w, w}, h[j], a);
        }
        y[o] = a;
    }
    constexpr int VEC = 16 / sizeof(T);
#pragma unroll
    for (int px = 0; px < 2; ++px) {
        if (px == 1 && i1 == i) break;
#pragma unroll
        for (int q = 0; q < 16 / VEC; ++q) {
            Vec16<T> v;
#pragma unroll
            for (int e = 0; e < VEC; ++e) v.v[e] = from_f32<T>(y[q * VEC + e][px]);
            *reinterpret_cast<Vec16<T>*>(out + (i + px) * 16 + q * VEC) = v;
        }
    }
}

}  // namespace s2m2

static int stem_mlp_impl(const void* x8, const float* w0, const float* b0, const float* w1, const float* b1, void* out, long long npix,
                             int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(x8 && w0 && b0 && w1 && b1 && out && npix > 0, "stem_mlp: bad arguments");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid = grid1((npix + 1) / 2);                                      // two pixels per thread
    if (dtype == S2M2_F16) hipLaunchKernelGGL((stem_mlp_kernel<half_t>), grid, dim3(256), 0, st, (const half_t*)x8, w0, b0, w1, b1, (half_t*)out, npix);
    else if (dtype == S2M2_F32) hipLaunchKernelGGL((stem_mlp_kernel<float>), grid, dim3(256), 0, st, (const float*)x8, w0, b0, w1, b1, (float*)out, npix);
    else return set_error("stem_mlp: unsupported dtype %d", dtype);
    return check_launch("stem_mlp");
}
extern "C" int s2m2_stem_mlp(const void* x8, const float* w0, const float* b0, const float* w1, const float* b1, void* out, long long npix,
                             int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_stem_mlp", &stem_mlp_impl, stream, x8, w0, b0, w1, b1, out, npix, dtype);
}


static int tanh_impl(const void* x, void* y, long long n, int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(x && y && n > 0 && n % 8 == 0, "tanh: bad arguments (n must be a multiple of 8)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == S2M2_F16) hipLaunchKernelGGL((tanh_kernel<half_t>), grid1(n / 8), dim3(256), 0, st, (const half_t*)x, (half_t*)y, n / 8);
    else if (dtype == S2M2_F32) hipLaunchKernelGGL((tanh_kernel<float>), grid1(n / 4), dim3(256), 0, st, (const float*)x, (float*)y, n / 4);
    else return set_error("tanh: unsupported dtype %d", dtype);
    return check_launch("tanh");
}
extern "C" int s2m2_tanh(const void* x, void* y, long long n, int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_tanh", &tanh_impl, stream, x, y, n, dtype);
}


// ---------------------------------------------------------------------------------------------------------------
// image_pad (reference src/s2m2/core/utils/image_utils.py:27-71): pad (B,C,H,W) to multiples of `factor` -- the border is NOT
// zero: it is a bilinear (align_corners=False) upsampling of the adaptive average pooling of the ZERO-padded image to
// (H/factor, W/factor), and the original image is pasted back in the middle.  Two small launches: pooling, then fill + paste.
// ---------------------------------------------------------------------------------------------------------------
namespace s2m2 {

template <typename TI>
__global__ __launch_bounds__(256) void pad_pool_kernel(const TI* __restrict__ img, float* __restrict__ pooled, int BC, int H, int W,
                                                       int Hn, int Wn, int Ho, int Wo, int hs, int ws) {
    // one wave per output bin: adaptive_avg_pool2d bins [floor(i*Hn/Ho), ceil((i+1)*Hn/Ho)) of the zero-padded image
    const int lane = threadIdx.x & 63;
    const long long bin = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    if (bin >= (long long)BC * Ho * Wo) return;
    const int ox = (int)(bin % Wo);
    const long long t = bin / Wo;
    const int oy = (int)(t % Ho);
    const long long bc = t / Ho;
    const int y0 = (int)(((long long)oy * Hn) / Ho), y1 = (int)((((long long)oy + 1) * Hn + Ho - 1) / Ho);
    const int x0 = (int)(((long long)ox * Wn) / Wo), x1 = (int)((((long long)ox + 1) * Wn + Wo - 1) / Wo);
    const int bw = x1 - x0, n = (y1 - y0) * bw;
    const TI* src = img + bc * (long long)H * W;
    float s = 0.f;
    for (int k = lane; k < n; k += 64) {
        const int yy = y0 + k / bw - hs, xx = x0 + k % bw - ws;       // coordinates in the original image
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) s += (float)src[(long long)yy * W + xx];
    }
    s = wave_sum(s);
    if (lane == 0) pooled[bin] = s / (float)n;
}

template <typename TI>
__global__ __launch_bounds__(256) void pad_fill_kernel(const TI* __restrict__ img, const float* __restrict__ pooled, float* __restrict__ out,
                                                       int BC, int H, int W, int Hn, int Wn, int Ho, int Wo, int hs, int ws) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long long)BC * Hn * Wn) return;
    const int X = (int)(gid % Wn);
    const long long t = gid / Wn;
    const int Y = (int)(t % Hn);
    const long long bc = t / Hn;
    const int yy = Y - hs, xx = X - ws;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) { out[gid] = (float)img[(bc * H + yy) * (long long)W + xx]; return; }
    // F.interpolate(mode='bilinear', align_corners=False): src = max((dst + 0.5) * in/out - 0.5, 0)
    const float sy = fmaxf(((float)Y + 0.5f) * ((float)Ho / (float)Hn) - 0.5f, 0.f);
    const float sx = fmaxf(((float)X + 0.5f) * ((float)Wo / (float)Wn) - 0.5f, 0.f);
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < Ho - 1), x1 = x0 + (x0 < Wo - 1);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float* pp = pooled + bc * (long long)Ho * Wo;
    out[gid] = (1.f - ly) * ((1.f - lx) * pp[y0 * Wo + x0] + lx * pp[y0 * Wo + x1]) + ly * ((1.f - lx) * pp[y1 * Wo + x0] + lx * pp[y1 * Wo + x1]);
}

}  // namespace s2m2

static int image_pad_impl(const void* img, float* pooled, float* out, int B, int C, int H, int W, int factor, int img_dtype,
                              void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(img && pooled && out && B > 0 && C > 0 && factor > 0 && H >= factor && W >= factor, "image_pad: bad arguments");
    const int Hn = (H + factor - 1) / factor * factor, Wn = (W + factor - 1) / factor * factor;
    const int Ho = H / factor, Wo = W / factor;
    const int hs = (Hn - H) / 2, ws = (Wn - W) / 2;
    const int BC = B * C;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long long bins = (long long)BC * Ho * Wo, pix = (long long)BC * Hn * Wn;
    #define S2M2_PAD(TI) do { \
        hipLaunchKernelGGL((pad_pool_kernel<TI>), grid1(bins * 64), dim3(256), 0, st, (const TI*)img, pooled, BC, H, W, Hn, Wn, Ho, Wo, hs, ws); \
        hipLaunchKernelGGL((pad_fill_kernel<TI>), grid1(pix), dim3(256), 0, st, (const TI*)img, pooled, out, BC, H, W, Hn, Wn, Ho, Wo, hs, ws); } while (0)
    if (img_dtype == S2M2_F32) S2M2_PAD(float);
    else if (img_dtype == S2M2_F16) S2M2_PAD(half_t);
    else if (img_dtype == 2) S2M2_PAD(unsigned char);
    else return set_error("image_pad: unsupported image dtype %d", img_dtype);
    #undef S2M2_PAD
    return check_launch("image_pad");
}
extern "C" int s2m2_image_pad(const void* img, float* pooled, float* out, int B, int C, int H, int W, int factor, int img_dtype,
                              void* stream) {
    return s2m2::plan_dispatch("s2m2_image_pad", &image_pad_impl, stream, img, pooled, out, B, C, H, W, factor, img_dtype);
}

