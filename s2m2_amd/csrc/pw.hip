// K11 -- a 1x1 layer with ANY channel counts in the direct style of K9 / K10 (fp16): Conv2d(kernel 1) / Linear / ConvTranspose2d(2, stride 2)
// on up to four concatenated sources, bias, GELU / ReLU, optional pixel-shuffle store.
//
// Replaces the K5 v1 launches (conv.hip: both operands staged through LDS behind a block barrier per 64-byte K tile) of the layers the
// square-C direct form of K9 does not take:
//   LocalRefiner   corr_feat*.0 (32 -> 192, GELU), corr_feat*.2 (192 -> 128), conf_occ_feat.2 (64 -> 32),
//                  disp_corr_ctx_cat.0 (cat(96, 128, C, 32) -> 2C, GELU)                       refinenet.py:87-106,138-146
//   Unet / MRT     up_conv 2C -> C on the coarse grid                                          unet.py:32-37, stacked_MRT.py:29-34
//   mask heads     ConvTranspose2d(C -> 64 / 16, k 2, s 2), ConvTranspose2d(48 -> 9, k 1)      submodules.py:104-113,131-144
//
// One block = BM = 64 * NWM token rows x all Cout output channels.  A wave owns ONE 32-cout tile (wn) of one 64-row group (wm) and loads
// ITS weight fragments -- every k16 step of the layer, K / 16 <= 24 fragments of 1 KB, packed once on the host in MFMA-fragment order
// (pack.pw_frag) -- from global memory straight into MFMA operand registers at kernel entry, before the row tile is even requested: the
// weights of a block arrive under the latency of its activations, the K loop is KS x MT MFMAs without a barrier, a wait or an LDS write.
// Block barriers: one after the row tile is in LDS, one before the staging tile (which reuses that LDS) is written, one before it is read back.
#include <hip/hip_runtime.h>

#include "common.h"
#include "cv_lookup.h"
#include "epilogue.h"

namespace s2m2 {

struct PwArgs {
    const void* src[4];
    long long sstride[4];
    int c1, c2, c3;                     // cumulative channel counts: source 0 holds channels [0, c1), source 1 [c1, c2), 2 [c2, c3), 3 [c3, K)
    int K;                              // sum of the sources' channels (multiple of 8; the tile is zero beyond it)
    long long rows;
    const void* w;                      // fragment order: [cout tile][k16 step][lane] x 16 bytes
    const float* bias;
    const void* zero;
    void* out;
    long long out_stride;
    int Cout, act;
    int shuffle2, Ho, Wo;               // > 0: ConvTranspose2d(2, s 2) store, cout = (dy * 2 + dx) * shuffle2 + c', rows = N * Ho * Wo input pixels
};

template <int KS_, int NWN_, int NWM_>
struct PwCfg {
    static constexpr int KS = KS_, NWN = NWN_, NWM = NWM_, NW = NWN_ * NWM_, NT = 64 * NW;
    static constexpr int VEC = 8, K = 16 * KS, BM = 64 * NWM, CO = 32 * NWN;
    static constexpr int WM = 64, MT = 2, WN = 32, NTL = 1;        // (names used by stage_tile / CoutRegs)
    static constexpr int ARS = K + VEC, CRS = CO + VEC;            // LDS row strides (elements): 16 bytes of padding
    static constexpr int APR = K / VEC, A_IT = (BM * APR + NT - 1) / NT;
    static constexpr int CPR = CO / VEC, C_IT = (BM * CPR + NT - 1) / NT;
    static constexpr size_t LDS_BYTES = (size_t)BM * (ARS > CRS ? ARS : CRS) * sizeof(half_t);
    static_assert(KS >= 1 && KS <= 24 && NW >= 1 && NW <= 8 && LDS_BYTES <= 64 * 1024, "unsupported pointwise tile");
};

template <typename CFG>
__global__ __launch_bounds__(CFG::NT) void pw_direct_kernel(PwArgs p) {
    using T = half_t;
    constexpr int KS = CFG::KS, ARS = CFG::ARS, CRS = CFG::CRS, VEC = CFG::VEC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* A = reinterpret_cast<T*>(smem);                              // [BM][K + pad] row tile; later the staging tile [BM][CO + pad]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv / CFG::NWN, wn = wv - wm * CFG::NWN;
    const long long m0 = (long long)blockIdx.x * CFG::BM;

    // ---- 1. this wave's weight fragments, all of them (tracked loads: consumed by the fully unrolled K loop below)
    const raw16_t* wq = static_cast<const raw16_t*>(p.w) + (size_t)wn * KS * 64 + lane;
    raw16_t wr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) wr[s] = global_load16(wq + s * 64);
    CoutRegs<CFG> bias;
    bias.load(p.bias, p.zero, p.Cout, 0, wn, lane);

    // ---- 2. row tile: 16-byte pieces of up to four sources, concatenated along the channels, zero beyond K / past the last row
    raw16_t xr[CFG::A_IT];
#pragma unroll
    for (int it = 0; it < CFG::A_IT; ++it) {
        const int idx = tid + CFG::NT * it, row = idx / CFG::APR, pc = idx - row * CFG::APR;
        const int c = pc * VEC;
        const long long m = m0 + row;
        // (selects, no indexing of kernel-argument arrays with a runtime index: that would go through scratch)
        const T* base = static_cast<const T*>(c < p.c1 ? p.src[0] : c < p.c2 ? p.src[1] : c < p.c3 ? p.src[2] : p.src[3]);
        const long long st = c < p.c1 ? p.sstride[0] : c < p.c2 ? p.sstride[1] : c < p.c3 ? p.sstride[2] : p.sstride[3];
        const int c0 = c < p.c1 ? 0 : c < p.c2 ? p.c1 : c < p.c3 ? p.c2 : p.c3;
        const bool ok = idx < CFG::BM * CFG::APR && m < p.rows && c < p.K;
        xr[it] = global_load16(ok ? base + m * st + (c - c0) : static_cast<const T*>(p.zero));
    }
#pragma unroll
    for (int it = 0; it < CFG::A_IT; ++it) {
        const int idx = tid + CFG::NT * it, row = idx / CFG::APR, pc = idx - row * CFG::APR;
        if (idx < CFG::BM * CFG::APR) *reinterpret_cast<raw16_t*>(A + (size_t)row * ARS + pc * VEC) = xr[it];
    }
    __syncthreads();

    // ---- 3. K loop: D[cout][row] += W fragment . X fragment, KS k16 steps x 2 row tiles, no synchronisation
    float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    const T* arow = A + (size_t)(wm * 64 + l31) * ARS + hi * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        Frag<T> wf;
        wf.v = __builtin_bit_cast(half8_t, wr[s]);
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i) {
            Frag<T> xf;
            load_frag(xf, arow + (size_t)i * 32 * ARS + s * 16);
            mma32(acc[i][0], wf, xf);
        }
    }
    __syncthreads();                                                // every wave is done with the row tile: its LDS becomes the staging tile

    // ---- 4. bias + activation -> staging tile [row][cout] -> coalesced 16-byte stores (pixel shuffle for ConvTranspose 2x2 s2)
    switch (p.act) {                                                // block-uniform
        case S2M2_ACT_GELU: stage_tile<CFG, T, S2M2_ACT_GELU>(acc, A, bias, 1.0f, wm, wn, lane); break;
        case S2M2_ACT_RELU: stage_tile<CFG, T, S2M2_ACT_RELU>(acc, A, bias, 1.0f, wm, wn, lane); break;
        default: stage_tile<CFG, T, S2M2_ACT_NONE>(acc, A, bias, 1.0f, wm, wn, lane); break;
    }
    __syncthreads();
    T* outp = static_cast<T*>(p.out);
#pragma unroll
    for (int it = 0; it < CFG::C_IT; ++it) {
        const int idx = tid + CFG::NT * it, row = idx / CFG::CPR, pc = idx - row * CFG::CPR;
        const int co = pc * VEC;
        const long long m = m0 + row;
        if (idx >= CFG::BM * CFG::CPR || m >= p.rows || co >= p.Cout) continue;
        const raw16_t v = *reinterpret_cast<const raw16_t*>(A + (size_t)row * CRS + co);
        long long opix = m;
        int oc = co;
        if (p.shuffle2) {                                           // cout = (dy * 2 + dx) * C' + c' (same map as K5, conv.hip store_tile)
            const int sub = co / p.shuffle2;
            oc = co - sub * p.shuffle2;
            const int x = (int)(m % p.Wo);
            const long long t = m / p.Wo;
            const int y = (int)(t % p.Ho);
            const long long n = t / p.Ho;
            opix = (n * (2 * p.Ho) + 2 * y + (sub >> 1)) * (2LL * p.Wo) + 2 * x + (sub & 1);
        }
        *reinterpret_cast<raw16_t*>(outp + opix * p.out_stride + oc) = v;
    }
}

template <int KS, int NWN, int NWM>
static int launch_pw(const PwArgs& a, hipStream_t st) {
    using CFG = PwCfg<KS, NWN, NWM>;
    auto kern = pw_direct_kernel<CFG>;
    static size_t lds_granted[kMaxDevices] = {};
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted, "pw_direct")) return 1;
    const long long nblk = (a.rows + CFG::BM - 1) / CFG::BM;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(CFG::NT), CFG::LDS_BYTES, st, a);
    return check_launch("pw_direct");
}

// k16 steps the library is instantiated for (K rounded up to 16 must be one of them) and waves per block by cout tiles: up to 2 cout tiles
// -> four 64-row groups, up to 4 -> one row group ... (a block has 4 - 8 waves)
static int pw_ks(int K) {
    const int ks = (K + 15) / 16;
    switch (ks) { case 2: case 3: case 4: case 6: case 8: case 12: case 16: case 24: return ks; default: return 0; }
}

// 64-row groups per block: four for a single cout tile, two for 2 - 3 tiles (a block has 4 - 8 waves), fewer where the row tile
// (K + 8 halfs per row) would not fit 64 KB of LDS
constexpr int pw_nwm(int ks, int nwn) {
    int want = nwn == 1 ? 4 : (nwn <= 3 ? 2 : 1);
    const int rs = (16 * ks + 8) > (32 * nwn + 8) ? (16 * ks + 8) : (32 * nwn + 8);
    while (want > 1 && 64 * want * rs * 2 > 64 * 1024) want /= 2;
    return want;
}

template <int KS>
static int dispatch_pw_n(const PwArgs& a, hipStream_t st) {
    const int nwn = (a.Cout + 31) / 32;
    switch (nwn) {
        case 1: return launch_pw<KS, 1, pw_nwm(KS, 1)>(a, st);
        case 2: return launch_pw<KS, 2, pw_nwm(KS, 2)>(a, st);
        case 3: return launch_pw<KS, 3, pw_nwm(KS, 3)>(a, st);
        case 4: return launch_pw<KS, 4, 1>(a, st);
        case 6: return launch_pw<KS, 6, 1>(a, st);
        case 8: return launch_pw<KS, 8, 1>(a, st);
        default: return set_error("pw_direct: Cout=%d (32-cout tiles: 1, 2, 3, 4, 6 or 8)", a.Cout);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// K3 + corr_feat fused (fp16): the two-level cost-volume lookup of one refinement iteration, `corr / 16 -> 1x1 (9 -> 96) -> GELU ->
// 1x1 (96 -> 64)` for both levels (reference refinenet.py:87-96,138-141 on top of submodules.py:39-60), as ONE launch per iteration instead
// of K3 + two K11 launches: a block owns 64 pixels, computes their 2 x 9 taps into an LDS tile (the arithmetic of cv_lookup.h, rounded to
// fp16 exactly where K3 stored them), runs the two block-diagonal layers (32 -> 192 with GELU, 192 -> 128; the 1/16 is folded into the first
// weight by the engine) with the 192-wide hidden rows staying in LDS, and stores the 128 correlation features.  Bit-identical to the three
// launches it replaces (tests/test_hip_pw.py).  Six waves: one per 32-cout tile of the first layer, four of them own the tiles of the second.
// ---------------------------------------------------------------------------------------------------------------------------
struct CorrFeatArgs {
    const void* cv;                     // (B, h, w, pitch) fp16, volume rows `pitch` apart
    const float* disp;                  // (B, 1, h, w) fp32
    const void* wa;                     // pack.pw_frag of the (192, 32) first layer (taps of level 0 in columns 0..8, of level 1 in 16..24)
    const float* ba;                    // (192)
    const void* wb;                     // pack.pw_frag of the (128, 192) second layer
    const float* bb;                    // (128)
    const void* zero;
    void* out;                          // (B*h*w, out_stride) fp16, 128 channels written
    long long out_stride;
    void* corr;                         // optional (B*h*w, 32) fp16: the lookups themselves (parity captures), or null
    long long rows;                     // B * h * w
    int h, w, pitch;
};

struct CorrFeatS1 { static constexpr int MT = 2, NTL = 1, WM = 64, WN = 32, CRS = 192 + 8; };    // hidden tile [64][192 + pad]
struct CorrFeatS2 { static constexpr int MT = 2, NTL = 1, WM = 64, WN = 32, CRS = 128 + 8; };    // staging tile [64][128 + pad]

__global__ __launch_bounds__(384) void corr_feat_kernel(CorrFeatArgs p) {
    using T = half_t;
    constexpr int ARS = 32 + 8, HRS = CorrFeatS1::CRS, CRS = CorrFeatS2::CRS, KS1 = 2, KS2 = 12;
    __shared__ __attribute__((aligned(16))) T A[64 * ARS];          // lookups: 9 + 9 taps in 32 channels
    __shared__ __attribute__((aligned(16))) T H[64 * HRS];          // hidden rows; the staging tile of the output afterwards
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);       // 0..5: cout tile of the first layer; 0..3 also of the second
    const long long m0 = (long long)blockIdx.x * 64;

    // weights first: two fragments of layer one, twelve of layer two (waves 4, 5 have no second tile: they re-read tile 3, unused)
    raw16_t wa[KS1], wb[KS2];
    const raw16_t* qa = static_cast<const raw16_t*>(p.wa) + (size_t)wn * KS1 * 64 + lane;
    const raw16_t* qb = static_cast<const raw16_t*>(p.wb) + (size_t)(wn < 4 ? wn : 3) * KS2 * 64 + lane;
#pragma unroll
    for (int s = 0; s < KS1; ++s) wa[s] = global_load16(qa + s * 64);
#pragma unroll
    for (int s = 0; s < KS2; ++s) wb[s] = global_load16(qb + s * 64);
    CoutRegs<CorrFeatS1> ba;
    ba.load(p.ba, p.zero, 192, 0, wn, lane);
    CoutRegs<CorrFeatS2> bb;
    bb.load(p.bb, p.zero, 128, 0, wn < 4 ? wn : 3, lane);

    // ---- lookups: element (pixel r, channel c) of the tile; c = 16 * level + k for k < 9, zero elsewhere
    for (int idx = tid; idx < 64 * 32; idx += 384) {
        const int r = idx >> 5, c = idx & 31;
        const long long m = m0 + r;
        const int level = c >> 4, k = c & 15;
        float v = 0.f;
        if (k < 9 && m < p.rows) {
            const int i = (int)(m % p.w);
            const long long rowid = m / p.w;
            const T* img = static_cast<const T*>(p.cv) + (size_t)rowid * p.w * p.pitch;
            v = lookup_tap<T>(img, i, p.disp[m], level, k, 4, p.w, p.pitch);
        }
        A[r * ARS + c] = (T)v;
    }
    __syncthreads();
    if (p.corr != nullptr) {                                        // (block-uniform) the lookups themselves, for parity captures
        for (int idx = tid; idx < 64 * 4; idx += 384) {
            const int r = idx >> 2, pc = idx & 3;
            if (m0 + r < p.rows)
                *reinterpret_cast<raw16_t*>(static_cast<T*>(p.corr) + (m0 + r) * 32 + pc * 8) = *reinterpret_cast<const raw16_t*>(A + r * ARS + pc * 8);
        }
    }

    // ---- layer one: hidden[:, 32 wn .. + 32) = GELU(Wa . taps + ba)
    float16_t acc[2][1];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
#pragma unroll
    for (int s = 0; s < KS1; ++s) {
        Frag<T> wf;
        wf.v = __builtin_bit_cast(half8_t, wa[s]);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            Frag<T> xf;
            load_frag(xf, A + (size_t)(i * 32 + l31) * ARS + hi * 8 + s * 16);
            mma32(acc[i][0], wf, xf);
        }
    }
    stage_tile<CorrFeatS1, T, S2M2_ACT_GELU>(acc, H, ba, 1.0f, 0, wn, lane);
    __syncthreads();

    // ---- layer two: out[:, 32 wn .. + 32) = Wb . hidden + bb   (waves 0..3)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
    if (wn < 4) {
#pragma unroll
        for (int s = 0; s < KS2; ++s) {
            Frag<T> wf;
            wf.v = __builtin_bit_cast(half8_t, wb[s]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                Frag<T> xf;
                load_frag(xf, H + (size_t)(i * 32 + l31) * HRS + hi * 8 + s * 16);
                mma32(acc[i][0], wf, xf);
            }
        }
    }
    __syncthreads();                                                // every wave is done reading the hidden rows: H becomes the staging tile
    if (wn < 4) stage_tile<CorrFeatS2, T, S2M2_ACT_NONE>(acc, H, bb, 1.0f, 0, wn, lane);
    __syncthreads();
    T* outp = static_cast<T*>(p.out);
    for (int idx = tid; idx < 64 * 16; idx += 384) {
        const int r = idx >> 4, pc = idx & 15;
        if (m0 + r < p.rows) *reinterpret_cast<raw16_t*>(outp + (m0 + r) * p.out_stride + pc * 8) = *reinterpret_cast<const raw16_t*>(H + (size_t)r * CRS + pc * 8);
    }
}

}  // namespace s2m2

extern "C" int s2m2_corr_feat(const void* cv, const float* disp, const void* wa_frag, const float* ba, const void* wb_frag, const float* bb,
                              void* out, long long out_stride, void* corr_out, int B, int h, int w, int cv_pitch, int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(cv && disp && wa_frag && wb_frag && out, "corr_feat: null pointer");
    S2M2_REQUIRE(dtype == S2M2_F16, "corr_feat: fp16 only (the fp32 mode runs K3 and the two layers as separate launches)");
    S2M2_REQUIRE(B > 0 && h > 0 && w > 1 && w % 2 == 0, "corr_feat: bad shape B=%d h=%d w=%d (w even)", B, h, w);
    if (cv_pitch == 0) cv_pitch = w;
    S2M2_REQUIRE(cv_pitch >= w && out_stride >= 128 && out_stride % 8 == 0, "corr_feat: cv_pitch=%d must be at least w, out_stride=%lld a multiple of 8 and at least 128",
                 cv_pitch, out_stride);
    CorrFeatArgs a;
    a.cv = cv; a.disp = disp; a.wa = wa_frag; a.ba = ba; a.wb = wb_frag; a.bb = bb; a.out = out; a.out_stride = out_stride; a.corr = corr_out;
    a.rows = (long long)B * h * w; a.h = h; a.w = w; a.pitch = cv_pitch;
    a.zero = zero_page();
    S2M2_REQUIRE(a.zero, "corr_feat: cannot allocate the zero page");
    hipLaunchKernelGGL(corr_feat_kernel, dim3((unsigned)((a.rows + 63) / 64)), dim3(384), 0, static_cast<hipStream_t>(stream), a);
    return check_launch("corr_feat");
}

extern "C" int s2m2_pw_direct_supported(int K, int Cout, int dtype) {
    const int nwn = (Cout + 31) / 32;
    return dtype == S2M2_F16 && K > 0 && K % 8 == 0 && Cout > 0 && Cout % 8 == 0 && s2m2::pw_ks(K) != 0 &&
           (nwn == 1 || nwn == 2 || nwn == 3 || nwn == 4 || nwn == 6 || nwn == 8);
}

extern "C" int s2m2_pw_direct(const s2m2_pw_desc* d, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(d, "pw_direct: null descriptor");
    S2M2_REQUIRE(d->nsrc >= 1 && d->nsrc <= 4 && d->rows > 0 && d->rows < (1LL << 40), "pw_direct: nsrc=%d rows=%lld", d->nsrc, d->rows);
    S2M2_REQUIRE(d->weight_frag && d->out && d->out_stride % 8 == 0, "pw_direct: null weight / out, or an out_stride that is not a multiple of 8");
    int K = 0, cum[4] = {0, 0, 0, 0};
    for (int i = 0; i < 4; ++i) {
        if (i < d->nsrc) {
            S2M2_REQUIRE(d->src[i] && d->src_c[i] > 0 && d->src_c[i] % 8 == 0 && d->src_stride[i] >= d->src_c[i] && d->src_stride[i] % 8 == 0,
                         "pw_direct: source %d needs a pointer, a channel count and a row stride that are multiples of 8", i);
            K += d->src_c[i];
        }
        cum[i] = K;
    }
    S2M2_REQUIRE(s2m2_pw_direct_supported(K, d->Cout, d->dtype), "pw_direct: K=%d Cout=%d dtype=%d is not supported (ask s2m2_pw_direct_supported)", K, d->Cout, d->dtype);
    S2M2_REQUIRE(d->act == S2M2_ACT_NONE || d->act == S2M2_ACT_GELU || d->act == S2M2_ACT_RELU, "pw_direct: act=%d (NONE, GELU or RELU)", d->act);
    S2M2_REQUIRE(d->shuffle2 == 0 || (d->shuffle2 % 8 == 0 && d->Cout == 4 * d->shuffle2 && d->Ho > 0 && d->Wo > 0 && d->rows % ((long long)d->Ho * d->Wo) == 0),
                 "pw_direct: shuffle2=%d needs Cout = 4 * shuffle2 (a multiple of 8 channels per sub-pixel) and rows = N * Ho * Wo", d->shuffle2);
    S2M2_REQUIRE(d->out_stride >= (d->shuffle2 ? d->shuffle2 : d->Cout), "pw_direct: out_stride=%lld is smaller than the output channels", d->out_stride);
    PwArgs a;
    for (int i = 0; i < 4; ++i) {
        const int j = i < d->nsrc ? i : d->nsrc - 1;
        a.src[i] = d->src[j];
        a.sstride[i] = d->src_stride[j];
    }
    a.c1 = cum[0]; a.c2 = cum[1]; a.c3 = cum[2]; a.K = K;
    a.rows = d->rows; a.w = d->weight_frag; a.bias = d->bias; a.out = d->out; a.out_stride = d->out_stride;
    a.Cout = d->Cout; a.act = d->act; a.shuffle2 = d->shuffle2; a.Ho = d->Ho; a.Wo = d->Wo;
    a.zero = zero_page();
    S2M2_REQUIRE(a.zero, "pw_direct: cannot allocate the zero page");
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (pw_ks(K)) {
        case 2: return dispatch_pw_n<2>(a, st);
        case 3: return dispatch_pw_n<3>(a, st);
        case 4: return dispatch_pw_n<4>(a, st);
        case 6: return dispatch_pw_n<6>(a, st);
        case 8: return dispatch_pw_n<8>(a, st);
        case 12: return dispatch_pw_n<12>(a, st);
        case 16: return dispatch_pw_n<16>(a, st);
        default: return dispatch_pw_n<24>(a, st);
    }
}
