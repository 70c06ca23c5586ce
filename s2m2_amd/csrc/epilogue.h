// Epilogue math shared by the GEMM-shaped kernels (K5 conv.hip, K9 chain.hip): activations, folded pre-LayerNorm.
#pragma once
#include "common.h"

#ifndef S2M2_CONV_DBG
#define S2M2_CONV_DBG 0          // ablation switches of conv.hip (bit 16: skip bias/activation math)
#endif

namespace s2m2 {

// Activations in the epilogue run on all BM*BN accumulators, so they must be a handful of VALU ops each: libm's erff / tanhf
// (~100 instructions with divergent range splits) made the GELU epilogue as expensive as the whole K loop.
//  erf: Abramowitz-Stegun 7.1.26, |error| <= 1.5e-7 (below fp32 round-off of the surrounding arithmetic);
//  exp: v_exp_f32 (1 ulp);  sigmoid / tanh from it.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_erf(float x) {
    const float ax = fabsf(x);
    const float t = fast_rcp(__builtin_fmaf(0.3275911f, ax, 1.0f));
    float pl = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    pl = __builtin_fmaf(pl, t, 1.421413741f);
    pl = __builtin_fmaf(pl, t, -0.284496736f);
    pl = __builtin_fmaf(pl, t, 0.254829592f);
    const float r = 1.0f - pl * t * fast_exp(-ax * ax);
    return copysignf(r, x);
}
// exact (erf) GELU = x * Phi(x) written as relu(x) - |x| * Phi(-|x|), Phi(-a) = erfc(a / sqrt 2) / 2 = poly(t) * t * exp(-a^2 / 2) with the
// same A&S 7.1.26 polynomial (t = 1 / (1 + p a / sqrt 2), the 1/2 and the 1/sqrt 2 folded into the constants): 12 plain operations +
// rcp + exp2 per element instead of 19 + 2 for 0.5 x (1 + erf(x / sqrt 2)) -- the GEMM epilogues evaluate it on every accumulator and
// are VALU-bound there (K10: 512 activations per pixel).  |error| <= 0.75e-7 |x|.
__device__ __forceinline__ float fast_gelu(float x) {
    const float ax = fabsf(x);
    const float t = fast_rcp(__builtin_fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    float pl = __builtin_fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
    pl = __builtin_fmaf(pl, t, 0.5f * 1.421413741f);
    pl = __builtin_fmaf(pl, t, 0.5f * -0.284496736f);
    pl = __builtin_fmaf(pl, t, 0.5f * 0.254829592f);
    const float q = pl * t * __builtin_amdgcn_exp2f((ax * ax) * (-0.5f * 1.44269504088896340736f));
    return __builtin_fmaf(-ax, q, fmaxf(x, 0.f));
}
// GELU whose result is rounded to fp16 (every GELU of the fp16 forward: the hidden activations feed the next layer's MFMAs as fp16):
//   x Phi(x) = relu(x) - a Phi(-a),  a = |x|,  Phi(-a) = 2^P(a),  P = the degree-7 minimax fit of log2 Phi(-a) on [0, 6.5] (Remez,
//   |P - log2 Phi(-a)| <= 6.8e-6: RELATIVE error 4.7e-6 of the a Phi(-a) term, absolute <= 8.1e-7 over all x; past 6.5 the leading
//   coefficient keeps P falling and the term is < 1e-13).
// Per PAIR of elements: 2 and + 7 v_pk_fma_f32 + 2 v_exp_f32 + 2 max + 1 v_pk_fma_f32 and NO reciprocal; fast_gelu costs 19 plain issues + 2 rcp
// + 2 exp2 per pair, and the transcendental unit runs at a quarter of the fma rate (~140 -> ~80 issue cycles per pair).
// Rounded to fp16 the result differs from the correctly rounded x Phi(x) at 0.13 % of a dense sweep of [-8, 8] (by one ulp); fast_gelu,
// whose error is absolute (0.75e-7 |x|), differs at 4.1 % -- all of them in the negative tail where the result is an fp16 subnormal.
// fp32 destinations keep fast_gelu (absolute error 3.3e-7 against 8.1e-7).
#ifndef S2M2_GELU16_POLY
#define S2M2_GELU16_POLY 1          // 0: A/B build with fast_gelu everywhere (the form up to round 4)
#endif
typedef float float2_t __attribute__((ext_vector_type(2)));
// two elements at a time, written on 2-vectors so that every fma is a v_pk_fma_f32 (left to the compiler, the |x| of a scalar form is folded
// into the fma as a source modifier, which the packed instruction does not have: seven unpacked fmas per element).  The empty asm pins the
// fp32 result in registers: without it the compiler fuses the last fma and the conversion to fp16 into v_fma_mixlo_f16 (one rounding
// instead of two) in SOME instantiations, and the forms of one layer (row-major / direct) stop agreeing bit for bit.
__device__ __forceinline__ float2_t fast_gelu16x2(float2_t x) {
    const float2_t a = __builtin_elementwise_abs(x);
    float2_t p = __builtin_elementwise_fma((float2_t)(-1.7577767721377313e-06f), a, (float2_t)(5.9936584875686094e-05f));
    p = __builtin_elementwise_fma(p, a, (float2_t)(-0.0009163629147224128f));
    p = __builtin_elementwise_fma(p, a, (float2_t)(0.008447385393083096f));
    p = __builtin_elementwise_fma(p, a, (float2_t)(-0.05382449924945831f));
    p = __builtin_elementwise_fma(p, a, (float2_t)(-0.4586157202720642f));
    p = __builtin_elementwise_fma(p, a, (float2_t)(-1.1511801481246948f));
    p = __builtin_elementwise_fma(p, a, (float2_t)(-1.0000038146972656f));
    const float2_t e = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
    float2_t r = __builtin_elementwise_fma(-a, e, __builtin_elementwise_max(x, (float2_t)(0.f)));
    asm volatile("" : "+v"(r));
    return r;
}
__device__ __forceinline__ float fast_gelu16(float x) {
    const float2_t r = fast_gelu16x2((float2_t){x, x});
    return r.x;
}
template <int ACT> __device__ __forceinline__ float activate(float x) {
#if defined(S2M2_GELU_ERF_FORM)                                  // A/B build: the round-1 form 0.5 x (1 + erf(x / sqrt 2))
    if (ACT == S2M2_ACT_GELU) return 0.5f * x * (1.0f + fast_erf(x * 0.70710678118654752440f));
#endif
    if (ACT == S2M2_ACT_GELU) return fast_gelu(x);
    if (ACT == S2M2_ACT_RELU) return fmaxf(x, 0.f);
    if (ACT == S2M2_ACT_SIGMOID) return fast_rcp(1.0f + fast_exp(-x));
    if (ACT == S2M2_ACT_TANH) return 1.0f - 2.0f * fast_rcp(1.0f + fast_exp(2.0f * x));
    return x;
}
// the activation of a value that is stored as T next
template <int ACT, typename T> __device__ __forceinline__ float activate_to(float x) {
    if constexpr (ACT == S2M2_ACT_GELU && sizeof(T) == 2 && S2M2_GELU16_POLY) return fast_gelu16(x);
    else return activate<ACT>(x);
}

// Pre-LayerNorm folded into a 1x1 layer (reference attentions.py:117,148,182,213,243: LayerNorm without affine feeding a Linear):
//   W . ((x - mean) * rstd) + b  =  rstd * (W . x  -  mean * rowsum(W)) + b
// The GEMM runs on the raw rows; mean / rstd of a row come from the A fragments the wave reads anyway (a lane owns one pixel
// and half of every k16 step: sum and sum of squares in fp32, one cross-half exchange at the end); rowsum(W) is packed once.
struct LnRow { float mean, rstd; };

// fp16 rows: products and sums are exact-ish in fp32 (error ~1e-7 * (mean/std)^2 relative to the variance, far below the fp16
// rounding of the operands).  fp32 rows: sums are taken about the row's first element, so a mean much larger than the spread
// does not cancel in q/C - mean^2.
__device__ __forceinline__ void ln_accumulate(const Frag<half_t>& f, float& s, float& q, float) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 one = {(_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const h2 v = {f.v[2 * e], f.v[2 * e + 1]};
        s = __builtin_amdgcn_fdot2(v, one, s, false);
        q = __builtin_amdgcn_fdot2(v, v, q, false);
    }
}
__device__ __forceinline__ void ln_accumulate(const Frag<float>& f, float& s, float& q, float shift) {
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = f.v[e] - shift; s += d; q = __builtin_fmaf(d, d, q); }
}

// Per-output-channel fp32 vectors (bias, rowsum(W)) in the accumulator layout of one wave: quad g of MFMA tile j holds couts
// wn*WN + j*32 + 8g + 4hi .. +3.  They are requested BEFORE the K loop with unconditional loads (out-of-range quads and a null
// source read the zero page): conditional loads inside the epilogue compile to one `s_waitcnt vmcnt(0)` per quad, i.e. 4*NTL
// serial memory latencies at the end of every block -- on the short layers that was a third of the kernel.
template <typename CFG>
struct CoutRegs {
    raw16_t v[CFG::NTL][4];
    __device__ __forceinline__ void load(const float* src, const void* zero, int Cout, int n0, int wn, int lane) {
        const int hi = lane >> 5;
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = n0 + wn * CFG::WN + j * 32 + 8 * g + 4 * hi;
                const float* q = (src != nullptr && co < Cout) ? src + co : static_cast<const float*>(zero);
                v[j][g] = global_load16(q);
            }
    }
};

// epilogue 1: bias, activation, scale in registers -> staging tile Cs[pixel][cout]; ACT is a compile-time constant here (a
// runtime switch per element made the compiler evaluate every activation and select)
template <typename CFG, typename T, int ACT, bool LN = false>
__device__ __forceinline__ void stage_tile(const float16_t (&acc)[CFG::MT][CFG::NTL], T* Cs, const CoutRegs<CFG>& bias, float out_scale,
                                           int wm, int wn, int lane, const LnRow* ln = nullptr, const CoutRegs<CFG>* wsum = nullptr) {
    const int hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i) {
        T* crow = Cs + (size_t)(wm * CFG::WM + i * 32 + (lane & 31)) * CFG::CRS;
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * CFG::WN + j * 32 + 8 * g + 4 * hi;         // local cout of the quad
                const raw16_t bv = bias.v[j][g];
                float v[4];
                if constexpr (LN) {
                    const raw16_t ws = wsum->v[j][g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(ln[i].rstd, __builtin_fmaf(-ln[i].mean, ws[e], acc[i][j][4 * g + e]), bv[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[i][j][4 * g + e] + bv[e];
                }
                if constexpr (ACT == S2M2_ACT_GELU && sizeof(T) == 2 && S2M2_GELU16_POLY) {
                    if (!(S2M2_CONV_DBG & 16)) {
                        const float2_t r0 = fast_gelu16x2((float2_t){v[0], v[1]}), r1 = fast_gelu16x2((float2_t){v[2], v[3]});
                        v[0] = r0.x * out_scale; v[1] = r0.y * out_scale; v[2] = r1.x * out_scale; v[3] = r1.y * out_scale;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (!LN && (S2M2_CONV_DBG & 16)) ? acc[i][j][4 * g + e] : activate<ACT>(v[e]) * out_scale;
                }
                if constexpr (sizeof(T) == 2) {
                    half4_t h = {from_f32<half_t>(v[0]), from_f32<half_t>(v[1]), from_f32<half_t>(v[2]), from_f32<half_t>(v[3])};
                    *reinterpret_cast<half4_t*>(crow + cl) = h;
                } else {
                    float4_t f = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<float4_t*>(crow + cl) = f;
                }
            }
        }
    }
}

// epilogue 2 helper: combine one staged 16-byte piece with the aux tensors (S2M2_EPI_*)
template <typename T>
__device__ __forceinline__ void aux_combine(Vec16<T>& v, int epi, const Vec16<T>& a0, const Vec16<T>& a1) {
#pragma unroll
    for (int e = 0; e < (int)(16 / sizeof(T)); ++e) {
        const float x = to_f32(v.v[e]), u = to_f32(a0.v[e]), w = to_f32(a1.v[e]);
        float o;
        if (epi == S2M2_EPI_ADD) o = x + u;
        else if (epi == S2M2_EPI_MUL) o = x * u;
        else if (epi == S2M2_EPI_GRU) o = (1.0f - u) * w + u * x;                                  // aux0 = z, aux1 = h, x = q
        else { const float gte = fminf(fmaxf(x, 0.01f), 0.99f); o = gte * u + (1.0f - gte) * w; }   // x = gate
        v.v[e] = from_f32<T>(o);
    }
}

}  // namespace s2m2
