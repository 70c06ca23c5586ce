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
#include "plan.h"
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
    if (nblk >= (1LL << 31)) return set_error("pw_direct: %lld row tiles do not fit a grid dimension", nblk);
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

// (measured and dropped in round 4, profiles/r04/ab_corr_feat.txt: the cost-volume lookup of a refinement iteration fused with its two corr_feat
// layers into one launch -- a block computing the 2 x 9 taps of its 64 pixels into the LDS tile itself, 32 -> 192 GELU -> 128 with the hidden
// rows resident -- was bit-identical to K3 + two K11 launches and SLOWER, 8.817 vs 8.754 ms per pair: 384 threads walking ~5 dependent tap
// fetches each are a longer latency chain than K3's one thread per tap over the whole image followed by two 12-14 us GEMM launches.)

}  // namespace s2m2

extern "C" int s2m2_pw_direct_supported(int K, int Cout, int dtype) {
    const int nwn = (Cout + 31) / 32;
    return dtype == S2M2_F16 && K > 0 && K % 8 == 0 && Cout > 0 && Cout % 8 == 0 && s2m2::pw_ks(K) != 0 &&
           (nwn == 1 || nwn == 2 || nwn == 3 || nwn == 4 || nwn == 6 || nwn == 8);
}

static int pw_direct_impl(const s2m2_pw_desc* d, void* stream) {
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
extern "C" int s2m2_pw_direct(const s2m2_pw_desc* d, void* stream) {
    return s2m2::plan_dispatch_desc<s2m2_pw_desc>("s2m2_pw_direct", &pw_direct_impl, d, stream);
}

