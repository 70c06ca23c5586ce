// K1 -- fused LayerNorm + all-pairs epipolar correlation (cost-volume construction).
//
// Replaces DispInit.forward's  layer_norm -> chunk -> einsum('...hic,...hjc->...hij')
// (/root/reference/src/s2m2/core/model/submodules.py:165,216-217).
//
// Decomposition (one image row (b,y) is an independent w x w x C contraction):
//   block  = one strip of TI = 32*NW left pixels of one image row  x  all w right pixels
//   wave   = 32 left pixels; its normalised A operand lives in registers for the whole block
//   loop   = chunks of TJ right pixels: raw tokens are fetched with 16-B coalesced loads (8 lanes per token),
//            LayerNorm'ed in fp32 (two-pass mean/var, 8-lane butterfly), rounded to T and staged in LDS
//            (row stride padded by 16 B -> conflict-free ds_read_b128 for the MFMA fragments), double buffered
//            so the HBM/L2 latency of chunk c+1 hides under the MFMAs and stores of chunk c;
//   store  = accumulators go through a wave-private LDS tile so that every lane writes 16 contiguous bytes
//            of one cost-volume row (a 32 x TJ tile = 32 segments of >= 128 B).
// Blocks of the same image row are made neighbours on one XCD (xcd_remap) so the right-image tokens, which
// every strip re-reads, are served by that XCD's L2.
//
// Algorithmic traffic per pair: 2*h*w*C*sizeof(T) read + h*w*w*sizeof(TO) written.
#include "common.h"

namespace s2m2 {

template <typename T, typename TO, int C_, int NW_, int TJ_, int NBUF_>
struct LnCorrCfg {
    static constexpr int C = C_, NW = NW_, TJ = TJ_, NBUF = NBUF_;
    static constexpr int VEC = 16 / sizeof(T);          // elements per 16-B piece
    static constexpr int PIECES = C / VEC;              // pieces per token
    static constexpr int LPT = 8;                       // lanes per token
    static constexpr int PPL = PIECES / LPT;            // pieces per lane per token
    static constexpr int TPR = NW * 64 / LPT;           // tokens per block-wide load round
    static constexpr int TI = NW * 32;                  // left pixels per block
    static constexpr int A_ROUNDS = TI / TPR;           // = 4
    static constexpr int B_ROUNDS = TJ / TPR;
    static constexpr int RS = C + VEC;                  // LDS row stride (elements): +16 B pad
    static constexpr int VECO = 16 / sizeof(TO);
    static constexpr int CRS = TJ + VECO;               // staging row stride (elements of TO)
    static constexpr int KSTEPS = C / 16;
    static constexpr int CT = TJ / 32;                  // 32-wide column tiles per chunk
    static constexpr size_t GB_BYTES = 2 * C * sizeof(float);
    static constexpr size_t B_BYTES = (size_t)NBUF * TJ * RS * sizeof(T);
    static constexpr size_t A_BYTES = (size_t)TI * RS * sizeof(T);
    static constexpr size_t CS_BYTES = (size_t)NW * 32 * CRS * sizeof(TO);
    static constexpr size_t AC_BYTES = A_BYTES > CS_BYTES ? A_BYTES : CS_BYTES;   // Cs aliases As (A is in registers)
    static constexpr size_t LDS_BYTES = GB_BYTES + B_BYTES + AC_BYTES;
    static_assert(PIECES % LPT == 0, "C must be a multiple of 8 pieces");
    static_assert(B_ROUNDS >= 1 && TJ % TPR == 0, "TJ must be a multiple of tokens-per-round");
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <typename CFG, typename T, int ROUNDS> struct RawTile { Vec16<T> p[ROUNDS][CFG::PPL]; };

// issue the 16-B loads of ROUNDS*TPR consecutive tokens starting at token t0 (clamped to the row)
template <typename CFG, typename T, int ROUNDS>
__device__ __forceinline__ void load_raw(RawTile<CFG, T, ROUNDS>& raw, const T* __restrict__ src, int t0, int w, int tid) {
    const int sub = tid & (CFG::LPT - 1);
    const int trow = tid / CFG::LPT;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        int tok = t0 + r * CFG::TPR + trow;
        tok = tok < w ? tok : w - 1;
        const T* p = src + (size_t)tok * CFG::C;
#pragma unroll
        for (int q = 0; q < CFG::PPL; ++q)
            raw.p[r][q] = *reinterpret_cast<const Vec16<T>*>(p + (sub + CFG::LPT * q) * CFG::VEC);
    }
}

// LayerNorm (eps 1e-5, biased variance, affine) each token in fp32 and write it, rounded to T, to dst[token][RS]
template <typename CFG, typename T, int ROUNDS>
__device__ __forceinline__ void normalize_store(const RawTile<CFG, T, ROUNDS>& raw, T* __restrict__ dst,
                                                const float* __restrict__ gb, int tid) {
    const int sub = tid & (CFG::LPT - 1);
    const int trow = tid / CFG::LPT;
    constexpr float inv_c = 1.0f / CFG::C;
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        float x[CFG::PPL][CFG::VEC];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < CFG::PPL; ++q)
#pragma unroll
            for (int e = 0; e < CFG::VEC; ++e) { x[q][e] = to_f32(raw.p[r][q].v[e]); s += x[q][e]; }
        const float mean = group_sum<CFG::LPT>(s) * inv_c;
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < CFG::PPL; ++q)
#pragma unroll
            for (int e = 0; e < CFG::VEC; ++e) { x[q][e] -= mean; ss += x[q][e] * x[q][e]; }
        const float rstd = rsqrtf(group_sum<CFG::LPT>(ss) * inv_c + 1e-5f);
        T* drow = dst + (size_t)(r * CFG::TPR + trow) * CFG::RS;
#pragma unroll
        for (int q = 0; q < CFG::PPL; ++q) {
            const int c0 = (sub + CFG::LPT * q) * CFG::VEC;
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < CFG::VEC; ++e) o.v[e] = from_f32<T>(x[q][e] * rstd * gb[c0 + e] + gb[CFG::C + c0 + e]);
            *reinterpret_cast<Vec16<T>*>(drow + c0) = o;
        }
    }
}

template <typename CFG, typename T, typename TO>
__global__ __launch_bounds__(CFG::NW * 64) void ln_corr_kernel(const T* __restrict__ feat, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, TO* __restrict__ cv,
                                                              int B, int h, int w, int nstrip) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gb = reinterpret_cast<float*>(smem);
    T* Bs = reinterpret_cast<T*>(smem + CFG::GB_BYTES);
    T* As = reinterpret_cast<T*>(smem + CFG::GB_BYTES + CFG::B_BYTES);
    TO* Cs = reinterpret_cast<TO*>(smem + CFG::GB_BYTES + CFG::B_BYTES);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = tid >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int row = bid / nstrip;            // (b, y)
    const int strip = bid - row * nstrip;
    const int b = row / h, y = row - b * h;
    const int i0 = strip * CFG::TI;
    const T* left = feat + ((size_t)(b * h + y) * w) * CFG::C;
    const T* right = feat + ((size_t)((B + b) * h + y) * w) * CFG::C;
    TO* cvrow = cv + (size_t)row * w * w;

    RawTile<CFG, T, CFG::A_ROUNDS> rawA;
    RawTile<CFG, T, CFG::B_ROUNDS> rawB;
    load_raw<CFG, T, CFG::A_ROUNDS>(rawA, left, i0, w, tid);
    load_raw<CFG, T, CFG::B_ROUNDS>(rawB, right, 0, w, tid);
    for (int c = tid; c < CFG::C; c += CFG::NW * 64) { gb[c] = gamma[c]; gb[CFG::C + c] = beta[c]; }
    __syncthreads();
    normalize_store<CFG, T, CFG::A_ROUNDS>(rawA, As, gb, tid);
    __syncthreads();

    // this wave's A operand: rows i0 + 32*wv + (lane&31), all C channels, as KSTEPS k16-fragments
    Frag<T> afrag[CFG::KSTEPS];
    {
        const T* ap = As + (size_t)(wv * 32 + (lane & 31)) * CFG::RS + (lane >> 5) * 8;
#pragma unroll
        for (int kk = 0; kk < CFG::KSTEPS; ++kk) load_frag(afrag[kk], ap + kk * 16);
    }
    __syncthreads();                          // As is dead from here on; its space becomes the store staging Cs

    const bool wave_active = (i0 + wv * 32) < w;
    const int nchunks = (w + CFG::TJ - 1) / CFG::TJ;
    TO* cs = Cs + (size_t)wv * 32 * CFG::CRS;

    for (int c = 0; c < nchunks; ++c) {
        T* bs = Bs + (size_t)(CFG::NBUF == 2 ? (c & 1) : 0) * CFG::TJ * CFG::RS;
        if (CFG::NBUF == 1 && c > 0) __syncthreads();            // previous chunk's readers are done
        normalize_store<CFG, T, CFG::B_ROUNDS>(rawB, bs, gb, tid);
        if (c + 1 < nchunks) load_raw<CFG, T, CFG::B_ROUNDS>(rawB, right, (c + 1) * CFG::TJ, w, tid);
        __syncthreads();
        if (!wave_active) continue;

        float16_t acc[CFG::CT];
#pragma unroll
        for (int ct = 0; ct < CFG::CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const T* bp = bs + (size_t)(lane & 31) * CFG::RS + (lane >> 5) * 8;
#pragma unroll
        for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
#pragma unroll
            for (int ct = 0; ct < CFG::CT; ++ct) {
                Frag<T> bf;
                load_frag(bf, bp + (size_t)ct * 32 * CFG::RS + kk * 16);
                mma32(acc[ct], afrag[kk], bf);
            }
        }
        // accumulators -> wave-private LDS tile -> 16-B row-contiguous global stores
#pragma unroll
        for (int ct = 0; ct < CFG::CT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                cs[acc_row(r, lane) * CFG::CRS + ct * 32 + (lane & 31)] = from_f32<TO>(acc[ct][r]);
        __builtin_amdgcn_wave_barrier();
        constexpr int PPR = CFG::TJ / CFG::VECO;          // 16-B pieces per staged row
        constexpr int ITERS = 32 * PPR / 64;
        const int j0 = c * CFG::TJ;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int q = it * 64 + lane;
            const int rr = q / PPR, pc = q - rr * PPR;
            const int i = i0 + wv * 32 + rr;
            const int j = j0 + pc * CFG::VECO;
            const Vec16<TO> v = *reinterpret_cast<const Vec16<TO>*>(cs + rr * CFG::CRS + pc * CFG::VECO);
            if (i < w && j < w) *reinterpret_cast<Vec16<TO>*>(cvrow + (size_t)i * w + j) = v;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
template <typename CFG, typename T, typename TO>
static int launch_ln_corr(const void* feat, const float* g, const float* bta, void* cv, int B, int h, int w, hipStream_t st) {
    auto kern = ln_corr_kernel<CFG, T, TO>;
    static bool attr_done = false;                       // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)CFG::LDS_BYTES) != hipSuccess)
            return set_error("ln_corr: cannot reserve %zu bytes of LDS", (size_t)CFG::LDS_BYTES);
        attr_done = true;
    }
    const int nstrip = (w + CFG::TI - 1) / CFG::TI;
    const int nblocks = B * h * nstrip;
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(CFG::NW * 64), CFG::LDS_BYTES, st,
                       static_cast<const T*>(feat), g, bta, static_cast<TO*>(cv), B, h, w, nstrip);
    return check_launch("ln_corr");
}

// tile configuration per (dtype, C): NW waves (32 left pixels each), TJ right pixels per chunk, NBUF LDS buffers
template <typename T, int C> struct LnCorrPick;
template <> struct LnCorrPick<half_t, 64>  { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 64, 4, 64, 2>; };
template <> struct LnCorrPick<half_t, 128> { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 128, 4, 64, 2>; };
template <> struct LnCorrPick<half_t, 192> { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 192, 4, 64, 2>; };
template <> struct LnCorrPick<half_t, 256> { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 256, 4, 64, 2>; };
template <> struct LnCorrPick<half_t, 384> { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 384, 4, 32, 2>; };
template <> struct LnCorrPick<float, 64>   { template <typename TO> using cfg = LnCorrCfg<float, TO, 64, 4, 32, 2>; };
template <> struct LnCorrPick<float, 128>  { template <typename TO> using cfg = LnCorrCfg<float, TO, 128, 4, 32, 2>; };
template <> struct LnCorrPick<float, 192>  { template <typename TO> using cfg = LnCorrCfg<float, TO, 192, 4, 32, 2>; };
template <> struct LnCorrPick<float, 256>  { template <typename TO> using cfg = LnCorrCfg<float, TO, 256, 2, 32, 2>; };
template <> struct LnCorrPick<float, 384>  { template <typename TO> using cfg = LnCorrCfg<float, TO, 384, 2, 32, 1>; };

template <typename T, typename TO>
static int dispatch_c(const void* feat, const float* g, const float* bta, void* cv, int B, int h, int w, int C, hipStream_t st) {
    switch (C) {
#define S2M2_CASE(CC) case CC: return launch_ln_corr<typename LnCorrPick<T, CC>::template cfg<TO>, T, TO>(feat, g, bta, cv, B, h, w, st);
        S2M2_CASE(64) S2M2_CASE(128) S2M2_CASE(192) S2M2_CASE(256) S2M2_CASE(384)
#undef S2M2_CASE
        default: return set_error("ln_corr: unsupported channel count C=%d (supported: 64,128,192,256,384)", C);
    }
}

}  // namespace s2m2

extern "C" int s2m2_ln_corr(const void* feat, const float* ln_w, const float* ln_b, void* cv, int B, int h, int w, int C,
                            int feat_dtype, int cv_dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(feat && ln_w && ln_b && cv, "ln_corr: null pointer");
    S2M2_REQUIRE(B > 0 && h > 0 && w > 0, "ln_corr: bad shape B=%d h=%d w=%d", B, h, w);
    S2M2_REQUIRE(w % 8 == 0, "ln_corr: w=%d must be a multiple of 8 (image width multiple of 32)", w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (feat_dtype == S2M2_F16 && cv_dtype == S2M2_F16) return dispatch_c<half_t, half_t>(feat, ln_w, ln_b, cv, B, h, w, C, st);
    if (feat_dtype == S2M2_F16 && cv_dtype == S2M2_F32) return dispatch_c<half_t, float>(feat, ln_w, ln_b, cv, B, h, w, C, st);
    if (feat_dtype == S2M2_F32 && cv_dtype == S2M2_F32) return dispatch_c<float, float>(feat, ln_w, ln_b, cv, B, h, w, C, st);
    return set_error("ln_corr: unsupported dtype pair feat=%d cv=%d", feat_dtype, cv_dtype);
}

extern "C" const char* s2m2_ln_corr_kernel_name(int C, int feat_dtype, int cv_dtype) {
    (void)C; (void)feat_dtype; (void)cv_dtype;
    return "ln_corr_kernel";
}
