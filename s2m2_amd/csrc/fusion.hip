// K10 -- FeatureFusion with 1x1 kernels as ONE launch (s2m2_feature_fusion).
//
// Reference feature_fusion.py:4-33 (kernel_size 1: every FeatureFusion of Unet / MRT except the 3x3 one of the CNN pyramid):
//     z = cat(z0, z1)                                   2C channels
//     h = GELU(W1 . z + b1)                             3C channels: rows [0, C) = feature_gate.0, rows [C, 3C) = feature_fusion.0
//     g = clamp(sigmoid(Wg . h[:C] + bg), .01, .99)     feature_gate.2
//     out = (Wf . h[C:] + bf) + g * z0 + (1 - g) * z1   feature_fusion.2
// As K5 launches this is two GEMMs with a 3C-wide intermediate going through HBM (K = 256 / 384: a handful of K tiles per block,
// overhead-dominated) -- 1.3 ms of the 10.4 ms pair.  Here a block owns BM rows:
//   * the z0 | z1 row tile stays in LDS for the whole kernel (A operand of the first layer AND the two mix operands of the epilogue);
//   * h is produced one C-wide slice at a time into a [BM][C+pad] LDS tile and consumed immediately as the next K range of the
//     second layer (slice 0 -> the gate accumulators, slices 1, 2 -> the fusion accumulators), so only BM x C of it ever exists;
//   * all weights form ONE stream of [C couts x 64 bytes of K] chunks (per slice: 2C/BK chunks of W1, then C/BK chunks of
//     [Wg | Wf]), D chunks in flight in registers as in K9;
//   * MFMA roles, staging and rounding points as in K5 / K9: h, g, the mix and the fusion term are each rounded to the I/O dtype
//     where the separate launches stored them.
#include "common.h"
#include "plan.h"
#include "epilogue.h"
#include <stdlib.h>
#include <type_traits>

namespace s2m2 {

struct FusionArgs {
    const void* z0;
    const void* z1;
    void* out;
    long long z0_stride, z1_stride, out_stride, rows;
    const void* w1;          // (3C, 2C)
    const void* w2;          // (C, 3C) = [Wg | Wf]
    const float* b1;         // 3C
    const float* bg;         // C
    const float* bf;         // C
    const void* zero;
    int up_h, up_w;          // > 0: z1 is the COARSE (N, up_h, up_w, C) tensor, read through the bilinear x2 resampling (nn.Upsample, align_corners=False)
};

template <typename T, int C_, int BM_, int NW_, int WP_ = 4>
struct FusionCfg {
    static constexpr int C = C_, BM = BM_, NW = NW_, NT = 64 * NW_;
    static constexpr int WP = WP_;                        // 16-byte pieces per weight row and chunk: 4 (64-byte K chunks) or 8 (128-byte)
    static constexpr int D = WP == 8 ? 2 : 4;             // chunks in flight (the same bytes either way)
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int BK = WP * VEC;                   // K elements per chunk
    static constexpr int RS = BK + VEC;                   // weight tile row stride in LDS
    static constexpr int KSTEPS = BK / 16;
    static constexpr int XRS = 2 * C + VEC;               // z0 | z1 tile row stride
    static constexpr int HRS = C + VEC;                   // h slice / staging tile row stride
    static constexpr int CRS = HRS;                       // (name used by stage_tile)
    static constexpr int WM = BM, MT = BM / 32, WN = C / NW, NTL = WN / 32;
    static constexpr int K0 = 2 * C / BK;                 // chunks of one first-layer slice
    static constexpr int K1 = C / BK;                     // chunks of one second-layer K range
    static constexpr int PER_SLICE = K0 + K1;
    static constexpr int TOTAL = 3 * PER_SLICE;
    static constexpr int WROWS = NT / WP;
    static constexpr int B_IT = C / WROWS;
    static constexpr int PPR = C / VEC;                   // 16-byte pieces per C-wide row
    static constexpr int X_IT = BM * PPR / NT;            // pieces per thread for one C-wide tile
    static constexpr size_t X_BYTES = (size_t)BM * XRS * sizeof(T);
    static constexpr size_t H_BYTES = (size_t)BM * HRS * sizeof(T);
    static constexpr size_t W_BYTES = (size_t)C * RS * sizeof(T);
    static constexpr size_t LDS_BYTES = X_BYTES + H_BYTES + 2 * W_BYTES;
    static_assert(C % 128 == 0 && K0 % D == 0 && K1 % D == 0 && (BM * PPR) % NT == 0 && BM % 32 == 0 && WN % 32 == 0 && C % WROWS == 0,
                  "unsupported fusion tile");
    static_assert(LDS_BYTES <= 160 * 1024, "fusion tile does not fit the 160 KB LDS");
};

// weight stream: position j -> slice s = j / PER_SLICE, r = j % PER_SLICE;  r < K0: rows [sC, sC+C) of W1, K chunk r;
// else rows [0, C) of W2, K chunk s*K1 + (r - K0)
template <typename CFG, typename T>
struct FusionStream {
    raw16_t r[CFG::D][CFG::B_IT];
    const T *w1, *w2;
    int lrow, pc;
    __device__ __forceinline__ void init(const FusionArgs& p, int tid) {
        w1 = static_cast<const T*>(p.w1); w2 = static_cast<const T*>(p.w2);
        lrow = tid / CFG::WP; pc = tid % CFG::WP;
    }
    __device__ __forceinline__ void fetch(int j, int SLOT) {      // j block-uniform, SLOT static after unrolling
        const int s = j / CFG::PER_SLICE, rr = j - s * CFG::PER_SLICE;
        const bool first = rr < CFG::K0;
        const T* base = first ? w1 + (size_t)s * CFG::C * (2 * CFG::C) + (size_t)rr * CFG::BK
                              : w2 + (size_t)(s * CFG::K1 + rr - CFG::K0) * CFG::BK;
        const int rstride = first ? 2 * CFG::C : 3 * CFG::C;
        const T* q = base + (size_t)lrow * rstride + pc * CFG::VEC;
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) r[SLOT][it] = global_load16(q + (size_t)it * CFG::WROWS * rstride);
    }
    __device__ __forceinline__ void stash(T* wb, int SLOT) const {
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it)
            *reinterpret_cast<raw16_t*>(wb + (size_t)(lrow + CFG::WROWS * it) * CFG::RS + pc * CFG::VEC) = r[SLOT][it];
    }
};

template <typename T> struct Quad;                                  // 4 consecutive channels of one row
template <> struct Quad<half_t> { half4_t v; };
template <> struct Quad<float> { float4_t v; };

// the z0 | z1 row tile of a block: 16-byte pieces requested into registers (z1 optionally through the x2 bilinear resampling)
template <typename CFG, typename T>
__device__ __forceinline__ void fusion_request_rows(const FusionArgs& p, long long m0, int tid, raw16_t (&xr0)[CFG::X_IT], raw16_t (&xr1)[CFG::X_IT]) {
    constexpr int VEC = CFG::VEC;
#pragma unroll
    for (int it = 0; it < CFG::X_IT; ++it) {
        const int idx = tid + CFG::NT * it, row = idx / CFG::PPR, pcx = idx - row * CFG::PPR;
        const long long m = m0 + row;
        const bool ok = m < p.rows;
        xr0[it] = global_load16(ok ? static_cast<const T*>(p.z0) + m * p.z0_stride + pcx * VEC : static_cast<const T*>(p.zero));
        if (p.up_h == 0) {
            xr1[it] = global_load16(ok ? static_cast<const T*>(p.z1) + m * p.z1_stride + pcx * VEC : static_cast<const T*>(p.zero));
        } else {
            // the x2 bilinear resampling of K7 (upsample.hip, same arithmetic and rounding) folded into the tile load: row m = (n, Y, X) of
            // the fine grid reads its four coarse neighbours
            const int Wo = 2 * p.up_w, Ho = 2 * p.up_h;
            const long long mm = ok ? m : 0;
            const int Xf = (int)(mm % Wo);
            const long long t = mm / Wo;
            const int Yf = (int)(t % Ho);
            const long long n = t / Ho;
            const float sy = fmaxf(((float)Yf + 0.5f) * 0.5f - 0.5f, 0.f), sx = fmaxf(((float)Xf + 0.5f) * 0.5f - 0.5f, 0.f);
            const int y0 = (int)sy, x0 = (int)sx;
            const int y1 = y0 + (y0 < p.up_h - 1), x1 = x0 + (x0 < p.up_w - 1);
            const float ly = sy - (float)y0, lx = sx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
            const T* zb = static_cast<const T*>(p.z1) + n * p.up_h * p.up_w * p.z1_stride + pcx * VEC;
            const Vec16<T> a = __builtin_bit_cast(Vec16<T>, global_load16(zb + ((long long)y0 * p.up_w + x0) * p.z1_stride));
            const Vec16<T> b = __builtin_bit_cast(Vec16<T>, global_load16(zb + ((long long)y0 * p.up_w + x1) * p.z1_stride));
            const Vec16<T> c = __builtin_bit_cast(Vec16<T>, global_load16(zb + ((long long)y1 * p.up_w + x0) * p.z1_stride));
            const Vec16<T> d = __builtin_bit_cast(Vec16<T>, global_load16(zb + ((long long)y1 * p.up_w + x1) * p.z1_stride));
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                o.v[e] = from_f32<T>(hy * (hx * to_f32(a.v[e]) + lx * to_f32(b.v[e])) + ly * (hx * to_f32(c.v[e]) + lx * to_f32(d.v[e])));
            xr1[it] = __builtin_bit_cast(raw16_t, o);
        }
    }
}
template <typename CFG, typename T>
__device__ __forceinline__ void fusion_stash_rows(T* X, int tid, const raw16_t (&xr0)[CFG::X_IT], const raw16_t (&xr1)[CFG::X_IT]) {
#pragma unroll
    for (int it = 0; it < CFG::X_IT; ++it) {
        const int idx = tid + CFG::NT * it, row = idx / CFG::PPR, pcx = idx - row * CFG::PPR;
        *reinterpret_cast<raw16_t*>(X + (size_t)row * CFG::XRS + pcx * CFG::VEC) = xr0[it];
        *reinterpret_cast<raw16_t*>(X + (size_t)row * CFG::XRS + CFG::C + pcx * CFG::VEC) = xr1[it];
    }
}

// out = round(round(accf + bf) + round(g * z0 + (1 - g) * z1)), g = clamp(round(sigmoid(accg + bg)), .01, .99): staged through H, stored coalesced
template <typename CFG, typename T>
__device__ __forceinline__ void fusion_mix_store(const FusionArgs& p, const float16_t (&accg)[CFG::MT][CFG::NTL], const float16_t (&accf)[CFG::MT][CFG::NTL],
                                                 const T* X, T* H, int tid, long long m0, const float* cv_lds = nullptr) {
    constexpr int C = CFG::C, VEC = CFG::VEC, XRS = CFG::XRS, HRS = CFG::HRS;
    const int lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    // ---- epilogue: out = round(round(accf + bf) + round(g * z0 + (1 - g) * z1)), g = clamp(round(sigmoid(accg + bg)), .01, .99)
    CoutRegs<CFG> bg, bf;
    if (cv_lds != nullptr) {                                         // (direct form: the per-cout vectors were copied to LDS by the prologue)
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int co = wn * CFG::WN + j * 32 + 8 * g + 4 * hi;
                bg.v[j][g] = *reinterpret_cast<const raw16_t*>(cv_lds + 3 * C + co);
                bf.v[j][g] = *reinterpret_cast<const raw16_t*>(cv_lds + 4 * C + co);
            }
    } else {
        bg.load(p.bg, p.zero, C, 0, wn, lane);
        bf.load(p.bf, p.zero, C, 0, wn, lane);
    }
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i) {
        const int row = i * 32 + l31;
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int cl = wn * CFG::WN + j * 32 + 8 * g4 + 4 * hi;
                const Quad<T> q0 = *reinterpret_cast<const Quad<T>*>(X + (size_t)row * XRS + cl);
                const Quad<T> q1 = *reinterpret_cast<const Quad<T>*>(X + (size_t)row * XRS + C + cl);
                Quad<T> o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float gate = to_f32(from_f32<T>(activate<S2M2_ACT_SIGMOID>(accg[i][j][4 * g4 + e] + bg.v[j][g4][e])));
                    const float gc = fminf(fmaxf(gate, 0.01f), 0.99f);
                    const float mix = to_f32(from_f32<T>(gc * to_f32(q0.v[e]) + (1.0f - gc) * to_f32(q1.v[e])));
                    const float fus = to_f32(from_f32<T>(accf[i][j][4 * g4 + e] + bf.v[j][g4][e]));
                    o.v[e] = from_f32<T>(fus + mix);
                }
                *reinterpret_cast<Quad<T>*>(H + (size_t)row * HRS + cl) = o;
            }
        }
    }
    __syncthreads();
    T* outp = static_cast<T*>(p.out);
#pragma unroll
    for (int it = 0; it < CFG::X_IT; ++it) {
        const int idx = tid + CFG::NT * it, row = idx / CFG::PPR, pcx = idx - row * CFG::PPR;
        const long long m = m0 + row;
        if (m < p.rows)
            *reinterpret_cast<raw16_t*>(outp + m * p.out_stride + pcx * VEC) = *reinterpret_cast<const raw16_t*>(H + (size_t)row * HRS + pcx * VEC);
    }
}

template <typename CFG, typename T>
__global__ __launch_bounds__(CFG::NT) void feature_fusion_kernel(FusionArgs p) {
    constexpr int C = CFG::C, BM = CFG::BM, BK = CFG::BK, RS = CFG::RS, XRS = CFG::XRS, HRS = CFG::HRS, D = CFG::D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* X = reinterpret_cast<T*>(smem);                                                   // [BM][2C + pad]: z0 | z1
    T* H = reinterpret_cast<T*>(smem + CFG::X_BYTES);                                    // [BM][C + pad]: one slice of h; staging tile at the end
    T* W0 = reinterpret_cast<T*>(smem + CFG::X_BYTES + CFG::H_BYTES);
    T* W1 = reinterpret_cast<T*>(smem + CFG::X_BYTES + CFG::H_BYTES + CFG::W_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const long long m0 = (long long)blockIdx.x * BM;

    FusionStream<CFG, T> ws;
    ws.init(p, tid);
    ws.fetch(0, 0);
    raw16_t xr0[CFG::X_IT], xr1[CFG::X_IT];
    fusion_request_rows<CFG, T>(p, m0, tid, xr0, xr1);
#pragma unroll
    for (int f = 1; f < D; ++f) ws.fetch(f, f);
    ws.stash(W0, 0);
    fusion_stash_rows<CFG, T>(X, tid, xr0, xr1);
    __syncthreads();

    float16_t accg[CFG::MT][CFG::NTL], accf[CFG::MT][CFG::NTL];                          // second layer: gate / fusion
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) accg[i][j][r] = accf[i][j][r] = 0.f;

    const int brow = (wn * CFG::WN + l31) * RS + hi * 8;
    // one K chunk: acc += Wtile(chunk) . A[:, k0 .. k0 + BK)
    auto chunk_step = [&](float16_t (&acc)[CFG::MT][CFG::NTL], const T* arow, int ars, int f, int j) __attribute__((always_inline)) {
        T* wb = (f & 1) ? W1 : W0;                                  // PER_SLICE, K0, K1 and D are even: LDS buffer of chunk j = j & 1 = f & 1
        T* wnext = (f & 1) ? W0 : W1;
        if (j + D < CFG::TOTAL) ws.fetch(j + D, f);
        const T* b = wb + brow;
#pragma unroll
        for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
            Frag<T> xf[CFG::MT], wf[CFG::NTL];
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], arow + (size_t)i * 32 * ars + kk * 16);
#pragma unroll
            for (int jn = 0; jn < CFG::NTL; ++jn) load_frag(wf[jn], b + (size_t)jn * 32 * RS + kk * 16);
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                for (int jn = 0; jn < CFG::NTL; ++jn) mma32(acc[i][jn], wf[jn], xf[i]);
        }
        if (j + 1 < CFG::TOTAL) ws.stash(wnext, (f + 1) % D);
        __syncthreads();
    };

#pragma unroll 1
    for (int s = 0; s < 3; ++s) {
        // ---- first layer, slice s: h[:, sC .. sC + C) = GELU(W1[sC .. sC + C, :] . z + b1)
        CoutRegs<CFG> b1;
        b1.load(p.b1 + s * C, p.zero, C, 0, wn, lane);
        float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int jbase = s * CFG::PER_SLICE;
        const T* xrow = X + (size_t)l31 * XRS + hi * 8;
#pragma unroll 1
        for (int c0 = 0; c0 < CFG::K0; c0 += D) {
#pragma unroll
            for (int f = 0; f < D; ++f) chunk_step(acc, xrow + (c0 + f) * BK, XRS, f, jbase + c0 + f);
        }
        stage_tile<CFG, T, S2M2_ACT_GELU>(acc, H, b1, 1.0f, 0, wn, lane);
        __syncthreads();
        // ---- second layer, K range s: gate (s == 0) or fusion (s == 1, 2) accumulators += [Wg | Wf][:, sC .. sC + C) . h slice
        const T* hrow = H + (size_t)l31 * HRS + hi * 8;
#pragma unroll 1
        for (int c0 = 0; c0 < CFG::K1; c0 += D) {
            if (s == 0) {
#pragma unroll
                for (int f = 0; f < D; ++f) chunk_step(accg, hrow + (c0 + f) * BK, HRS, f, jbase + CFG::K0 + c0 + f);
            } else {
#pragma unroll
                for (int f = 0; f < D; ++f) chunk_step(accf, hrow + (c0 + f) * BK, HRS, f, jbase + CFG::K0 + c0 + f);
            }
        }
        // (the trailing barrier of the last chunk: every wave is done with the h slice before the next one overwrites it)
    }

    fusion_mix_store<CFG, T>(p, accg, accf, X, H, tid, m0);
}

// ---------------------------------------------------------------------------------------------------------------
// K10, DIRECT form (fp16; weights in MFMA-fragment order, s2m2_feature_fusion_frag).  The kernel above stages every 64 / 128-byte K chunk
// of the weights through LDS behind a block barrier (36 - 72 barriers per block); on the short row counts of the coarse pyramid levels a
// block lives for the latency of that chain.  Here a wave owns 32 couts and reads ITS weight fragments -- one contiguous stream of
// 1 KB fragments in exactly the order it consumes them, packed once on the host (pack.fusion_frag) -- from global memory straight into
// MFMA operand registers, D fragments ahead; block barriers only where the h slice changes hands (6 per block).  Same arithmetic,
// k16 order and rounding points as the kernel above (tests/test_hip_fusion.py: bit-identical).
//   stream of wave tile t:  for slice s = 0, 1, 2:  W1[sC + 32t .. + 32, k16 steps 0 .. 2C/16),  then  [Wg | Wf][32t .. + 32, steps sC/16 .. (s+1)C/16)
// ---------------------------------------------------------------------------------------------------------------
// f(integral_constant<int, J0>), f(integral_constant<int, J0 + 1>), ... N times: a loop whose index is a constant expression inside the body
template <int J0, int N, typename F>
__device__ __forceinline__ void static_steps(F&& f) {
    if constexpr (N > 0) {
        f(std::integral_constant<int, J0>{});
        static_steps<J0 + 1, N - 1>(f);
    }
}

// (measured and dropped, profiles/r04/ab_minwaves.txt: three waves per SIMD asked of the allocator at C = 128 -- <= 168 registers with rings of
// 8 / 12 fragments instead of 132 + 96 / 156 + 48 with rings of 12 / 24 -- 8.793 vs 8.782 ms per pair; with 32-row tiles everywhere 8.768)
template <int C_, int BM_, int NW_, int D_>
struct FusionDirectCfg {
    static constexpr int C = C_, BM = BM_, NW = NW_, NT = 64 * NW_, D = D_;
    static constexpr int VEC = 8;
    static constexpr int XRS = 2 * C + VEC, HRS = C + VEC, CRS = HRS;
    static constexpr int WM = BM, MT = BM / 32, WN = C / NW, NTL = WN / 32;
    static constexpr int KS0 = 2 * C / 16, KS1 = C / 16, PER = KS0 + KS1, TOTAL = 3 * PER;    // k16 steps per slice: first layer, second layer
    static constexpr int PPR = C / VEC, X_IT = BM * PPR / NT;
    static constexpr size_t X_BYTES = (size_t)BM * XRS * sizeof(half_t);
    static constexpr size_t H_BYTES = (size_t)BM * HRS * sizeof(half_t);
    static constexpr size_t CV_BYTES = (size_t)5 * C * sizeof(float);                  // b1 (3C) | bg (C) | bf (C)
    static constexpr size_t LDS_BYTES = X_BYTES + H_BYTES + CV_BYTES;
    static_assert(NTL == 1 && BM % 32 == 0 && (BM * PPR) % NT == 0 && D <= PER && D * 4 <= 128 && 5 * C / 4 <= NT * 2, "unsupported direct fusion tile");
};

template <typename CFG>
__global__ __launch_bounds__(CFG::NT) void feature_fusion_direct_kernel(FusionArgs p) {
    using T = half_t;
    constexpr int C = CFG::C, BM = CFG::BM, XRS = CFG::XRS, HRS = CFG::HRS, D = CFG::D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* X = reinterpret_cast<T*>(smem);                              // [BM][2C + pad]: z0 | z1
    T* H = reinterpret_cast<T*>(smem + CFG::X_BYTES);               // [BM][C + pad]: one slice of h; staging tile at the end
    const int tid = threadIdx.x, lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const long long m0 = (long long)blockIdx.x * BM;

    // this wave's fragment stream (p.w1: the whole stream, w2 unused): position j at 16-byte slot (wn * TOTAL + j) * 64 + lane
    const raw16_t* wq = reinterpret_cast<const raw16_t*>(p.w1) + (size_t)wn * CFG::TOTAL * 64 + lane;
    // per-cout fp32 vectors -> LDS first (a load requested inside the K loops would be waited for with the whole ring ahead of it: loads
    // return in order), then the row tile, then the first D fragments
    float* Cv = reinterpret_cast<float*>(smem + CFG::X_BYTES + CFG::H_BYTES);            // b1 (3C) | bg (C) | bf (C)
    raw16_t cvr[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = tid + CFG::NT * it;                           // 16-byte piece of the 5C floats
        const float* src = q < 3 * C / 4 ? p.b1 + 4 * q : q < 4 * C / 4 ? p.bg + 4 * q - 3 * C : q < 5 * C / 4 ? p.bf + 4 * q - 4 * C
                                                                                                                 : static_cast<const float*>(p.zero);
        cvr[it] = global_load16(src);
    }
    raw16_t xr0[CFG::X_IT], xr1[CFG::X_IT];
    fusion_request_rows<CFG, T>(p, m0, tid, xr0, xr1);
    // the ring: untracked loads + counted waits (common.h: the compiler's own load tracking sinks the refills towards their use and the
    // prefetch distance collapses to one or two fragments -- seen in the ISA).  Fragment j is request number j; before step j at most
    // min(D, TOTAL - j) - 1 younger requests may be in flight.
    raw16_t ring[D];
#pragma unroll
    for (int f = 0; f < D; ++f) global_load16_async(ring[f], wq + f * 64);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int q = tid + CFG::NT * it;
        if (q < 5 * C / 4) *reinterpret_cast<raw16_t*>(Cv + 4 * q) = cvr[it];
    }
    fusion_stash_rows<CFG, T>(X, tid, xr0, xr1);
    __syncthreads();

    float16_t accg[CFG::MT][1], accf[CFG::MT][1];                    // second layer: gate / fusion
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accg[i][0][r] = accf[i][0][r] = 0.f;

    // one k16 step at stream position J: acc += fragment(J) . A[:, 16 k .. 16 k + 16), then the ring slot is refilled with fragment J + D
    auto step = [&](float16_t (&acc)[CFG::MT][1], const T* arow, int ars, auto jc) __attribute__((always_inline)) {
        constexpr int J = decltype(jc)::value;
        Frag<T> wf, xf[CFG::MT];
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], arow + (size_t)i * 32 * ars);
        wait_vmcnt<(J + D <= CFG::TOTAL ? D : CFG::TOTAL - J) - 1>();
        settle(ring[J % D]);
        wf.v = __builtin_bit_cast(half8_t, ring[J % D]);
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i) mma32(acc[i][0], wf, xf[i]);
        if constexpr (J + D < CFG::TOTAL) global_load16_async(ring[J % D], wq + (size_t)(J + D) * 64);
    };

    const T* xrow = X + (size_t)l31 * XRS + hi * 8;
    const T* hrow = H + (size_t)l31 * HRS + hi * 8;
    static_steps<0, 3>([&](auto sc) __attribute__((always_inline)) {
        constexpr int S = decltype(sc)::value;
        // ---- first layer, slice S: h[:, SC .. SC + C) = GELU(W1[SC .. SC + C, :] . z + b1)
        CoutRegs<CFG> b1;
#pragma unroll
        for (int g = 0; g < 4; ++g) b1.v[0][g] = *reinterpret_cast<const raw16_t*>(Cv + S * C + wn * CFG::WN + 8 * g + 4 * hi);
        float16_t acc[CFG::MT][1];
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
        static_steps<0, CFG::KS0>([&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            step(acc, xrow + K * 16, XRS, std::integral_constant<int, S * CFG::PER + K>{});
        });
        if constexpr (S > 0) __syncthreads();                       // every wave has read the previous h slice
        stage_tile<CFG, T, S2M2_ACT_GELU>(acc, H, b1, 1.0f, 0, wn, lane);
        __syncthreads();
        // ---- second layer, K range S: gate (S == 0) or fusion (S == 1, 2) accumulators += [Wg | Wf][:, SC .. SC + C) . h slice
        static_steps<0, CFG::KS1>([&](auto kc) __attribute__((always_inline)) {
            constexpr int K = decltype(kc)::value;
            if constexpr (S == 0) step(accg, hrow + K * 16, HRS, std::integral_constant<int, S * CFG::PER + CFG::KS0 + K>{});
            else step(accf, hrow + K * 16, HRS, std::integral_constant<int, S * CFG::PER + CFG::KS0 + K>{});
        });
    });
    __syncthreads();                                                // the last h slice has been read: H becomes the staging tile
    fusion_mix_store<CFG, T>(p, accg, accf, X, H, tid, m0, Cv);
}

template <int C, int BM, int NW, int D>
static int launch_fusion_direct(const FusionArgs& a, hipStream_t st) {
    using CFG = FusionDirectCfg<C, BM, NW, D>;
    auto kern = feature_fusion_direct_kernel<CFG>;
    static size_t lds_granted[kMaxDevices] = {};
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted, "feature_fusion")) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.rows + BM - 1) / BM)), dim3(CFG::NT), CFG::LDS_BYTES, st, a);
    return check_launch("feature_fusion");
}

template <typename T, int C, int BM, int NW, int WP = 4>
static int launch_fusion(const FusionArgs& a, hipStream_t st) {
    using CFG = FusionCfg<T, C, BM, NW, WP>;
    auto kern = feature_fusion_kernel<CFG, T>;
    static size_t lds_granted[kMaxDevices] = {};                     // per instantiation
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted, "feature_fusion")) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.rows + BM - 1) / BM)), dim3(CFG::NT), CFG::LDS_BYTES, st, a);
    return check_launch("feature_fusion");
}

}  // namespace s2m2

extern "C" int s2m2_feature_fusion_supported(int C, int dtype) {
    return (dtype == S2M2_F16 || dtype == S2M2_F32) && (C == 128 || C == 256);
}

// (C = 512, r06: the L model's 1/16 level, ten fusions per forward that ran as two K5 launches each; 16 waves per block, ring of 6 fragments: 8 spill)
extern "C" int s2m2_feature_fusion_frag_supported(int C, int dtype) {
    static const bool no512 = getenv("S2M2_FUSION_FRAG512") != nullptr && atoi(getenv("S2M2_FUSION_FRAG512")) == 0;        // A/B switch
    return dtype == S2M2_F16 && (C == 128 || C == 192 || C == 256 || C == 384 || (C == 512 && !no512));
}

static int feature_fusion_frag_impl(const void* z0, const void* z1, void* out, long long z0_stride, long long z1_stride, long long out_stride,
                                        long long rows, int C, const void* w_stream, const float* b1, const float* bg, const float* bf,
                                        int z1_coarse_h, int z1_coarse_w, int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(z0 && z1 && out && w_stream && b1 && bg && bf, "feature_fusion_frag: null pointer");
    S2M2_REQUIRE(s2m2_feature_fusion_frag_supported(C, dtype), "feature_fusion_frag: C=%d dtype=%d is not supported (fp16, C = 128, 192, 256, 384 or 512)", C, dtype);
    S2M2_REQUIRE(rows > 0 && rows < (1LL << 31), "feature_fusion_frag: rows=%lld", rows);
    S2M2_REQUIRE(z0_stride >= C && z1_stride >= C && out_stride >= C && z0_stride % 8 == 0 && z1_stride % 8 == 0 && out_stride % 8 == 0,
                 "feature_fusion_frag: row strides must be multiples of 8 and at least C");
    S2M2_REQUIRE((z1_coarse_h == 0 && z1_coarse_w == 0) || (z1_coarse_h > 0 && z1_coarse_w > 0 && rows % (4LL * z1_coarse_h * z1_coarse_w) == 0),
                 "feature_fusion_frag: rows=%lld is not a whole number of (2*%d) x (2*%d) images", rows, z1_coarse_h, z1_coarse_w);
    FusionArgs a;
    a.z0 = z0; a.z1 = z1; a.out = out; a.z0_stride = z0_stride; a.z1_stride = z1_stride; a.out_stride = out_stride; a.rows = rows;
    a.w1 = w_stream; a.w2 = nullptr; a.b1 = b1; a.bg = bg; a.bf = bf;
    a.up_h = z1_coarse_h; a.up_w = z1_coarse_w;
    a.zero = zero_page();
    S2M2_REQUIRE(a.zero, "feature_fusion_frag: cannot allocate the zero page");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // 32-row tiles while one round of them fits the chip, else 64-row tiles (half the weight traffic per row); S2M2_FUSION_DIRECT_BM forces one
    static const int force_bm = getenv("S2M2_FUSION_DIRECT_BM") ? atoi(getenv("S2M2_FUSION_DIRECT_BM")) : 0;
    const bool tall = force_bm ? force_bm == 64 : rows > (C == 128 ? 24576 : C == 192 ? 16384 : 8192);
    if (C == 128) return tall ? launch_fusion_direct<128, 64, 4, 12>(a, st) : launch_fusion_direct<128, 32, 4, 24>(a, st);
    if (C == 192) return tall ? launch_fusion_direct<192, 64, 6, 12>(a, st) : launch_fusion_direct<192, 32, 6, 24>(a, st);
    if (C == 384) return launch_fusion_direct<384, 32, 12, 16>(a, st);   // 12 waves: three per SIMD, <= 168 registers (64-row tiles spill)
    if (C == 512) return launch_fusion_direct<512, 32, 16, 6>(a, st);    // 16 waves: four per SIMD, <= 128 registers (a ring of 8 spills 6)
    return tall ? launch_fusion_direct<256, 64, 8, 16>(a, st) : launch_fusion_direct<256, 32, 8, 24>(a, st);
}
extern "C" int s2m2_feature_fusion_frag(const void* z0, const void* z1, void* out, long long z0_stride, long long z1_stride, long long out_stride,
                                        long long rows, int C, const void* w_stream, const float* b1, const float* bg, const float* bf,
                                        int z1_coarse_h, int z1_coarse_w, int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_feature_fusion_frag", &feature_fusion_frag_impl, stream, z0, z1, out, z0_stride, z1_stride, out_stride, rows, C, w_stream, b1, bg, bf, z1_coarse_h, z1_coarse_w, dtype);
}


static int feature_fusion_impl(const void* z0, const void* z1, void* out, long long z0_stride, long long z1_stride, long long out_stride,
                                   long long rows, int C, const void* w1, const float* b1, const void* w2, const float* bg,
                                   const float* bf, int z1_coarse_h, int z1_coarse_w, int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(z0 && z1 && out && w1 && w2 && b1 && bg && bf, "feature_fusion: null pointer");
    S2M2_REQUIRE(s2m2_feature_fusion_supported(C, dtype), "feature_fusion: C=%d dtype=%d is not supported (C = 128 or 256)", C, dtype);
    S2M2_REQUIRE(rows > 0 && rows < (1LL << 31), "feature_fusion: rows=%lld", rows);
    S2M2_REQUIRE(z0_stride >= C && z1_stride >= C && out_stride >= C && z0_stride % 8 == 0 && z1_stride % 8 == 0 && out_stride % 8 == 0,
                 "feature_fusion: row strides must be multiples of 8 and at least C");
    FusionArgs a;
    a.z0 = z0; a.z1 = z1; a.out = out; a.z0_stride = z0_stride; a.z1_stride = z1_stride; a.out_stride = out_stride; a.rows = rows;
    a.w1 = w1; a.w2 = w2; a.b1 = b1; a.bg = bg; a.bf = bf;
    a.up_h = z1_coarse_h; a.up_w = z1_coarse_w;
    S2M2_REQUIRE((z1_coarse_h == 0 && z1_coarse_w == 0) || (z1_coarse_h > 0 && z1_coarse_w > 0 && rows % (4LL * z1_coarse_h * z1_coarse_w) == 0),
                 "feature_fusion: rows=%lld is not a whole number of (2*%d) x (2*%d) images", rows, z1_coarse_h, z1_coarse_w);
    a.zero = zero_page();
    S2M2_REQUIRE(a.zero, "feature_fusion: cannot allocate the zero page");
    hipStream_t st = static_cast<hipStream_t>(stream);
    // measured end to end (same-box A/B): 32-row tiles (the 64-row tile of C = 128 needs 234 VGPRs for its three accumulator sets
    // and runs one block per CU); 128-byte K chunks (half the barriers of 64-byte chunks, +1.3 %) wherever the tile still fits
    static const bool narrow = getenv("S2M2_FUSION_CHUNK64") != nullptr;   // A/B switch
    static const int big = getenv("S2M2_FUSION_BM") ? atoi(getenv("S2M2_FUSION_BM")) : 0;   // experiment: 64- / 128-row tiles at C = 128
    if (dtype == S2M2_F16) {
        if (C == 128 && big == 128 && rows >= 32768) return launch_fusion<half_t, 128, 128, 4, 8>(a, st);
        if (C == 128 && big == 64 && rows >= 32768) return launch_fusion<half_t, 128, 64, 4, 8>(a, st);
        if (C == 128) return narrow ? launch_fusion<half_t, 128, 32, 4>(a, st) : launch_fusion<half_t, 128, 32, 4, 8>(a, st);
        if (rows > 8192) return launch_fusion<half_t, 256, 64, 8>(a, st);   // bulk rows: 64-row tiles (weights streamed once per 64 rows)
        return narrow ? launch_fusion<half_t, 256, 32, 8>(a, st) : launch_fusion<half_t, 256, 32, 8, 8>(a, st);
    }
    if (C == 128) return launch_fusion<float, 128, 32, 4>(a, st);
    return launch_fusion<float, 256, 32, 8>(a, st);
}
extern "C" int s2m2_feature_fusion(const void* z0, const void* z1, void* out, long long z0_stride, long long z1_stride, long long out_stride,
                                   long long rows, int C, const void* w1, const float* b1, const void* w2, const float* bg,
                                   const float* bf, int z1_coarse_h, int z1_coarse_w, int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_feature_fusion", &feature_fusion_impl, stream, z0, z1, out, z0_stride, z1_stride, out_stride, rows, C, w1, b1, w2, bg, bf, z1_coarse_h, z1_coarse_w, dtype);
}

