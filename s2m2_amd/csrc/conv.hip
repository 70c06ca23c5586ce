// K5 -- implicit-GEMM convolution / linear layer on MFMA with fused input concatenation and fused epilogues.
//
// Replaces every stride-1 nn.Conv2d / nn.ConvTranspose2d / nn.Linear call site of the hot path outside the CNN backbone
// (reference: refinenet.py:14-20,47-57,87-122; attentions.py:24-28,71-74,239-241,269-275; feature_fusion.py:15-21;
// stacked_MRT.py:22-34; unet.py:25-37; submodules.py:104-108,127-135; s2m2.py:65-67) together with the elementwise
// ops PyTorch runs as separate kernels around them: torch.cat of the inputs, bias add, GELU / ReLU / sigmoid / tanh,
// residual add, the ConvGRU gate arithmetic (refinenet.py:24-34) and the FeatureFusion gate mix (feature_fusion.py:24-31).
//
// GEMM view:  D[cout][pixel] = sum_k Wp[cout][k] * X[pixel][k],  k = (tap, channel) with channel fastest; X is gathered on
// the fly from up to four NHWC tensors (zero "same" padding), Wp is the weight packed once as (Cout, KH*KW*Cin).
//   block  = 256 threads = 4 waves, BM pixels x BN output channels, K swept in tiles of 128 bytes per row (64 fp16 / 32 fp32)
//            = 8 16-byte pieces; a piece never straddles a tap or a source (all channel counts are multiples of 8).
//   stage  = global -> registers (issued before the MFMAs of the current tile) -> LDS after them (double buffered, one
//            block barrier per K tile); LDS rows padded by 16 B (conflict-free ds_read_b128 fragment reads).
//   MFMA   = roles swapped as in K1: D = W_tile . X_tile^T, so a lane owns ONE pixel and 4 consecutive output channels
//            per register quad -> bias/activation in registers, quads go to an LDS staging tile, come back as 16-byte
//            pieces of whole pixel rows (NHWC: channels contiguous) for the aux epilogue and fully coalesced stores.
// fp16: v_mfma_f32_32x32x16_f16, fp32 accumulate.  fp32 (parity mode): exact v_mfma_f32_32x32x2_f32.
#include "common.h"

namespace s2m2 {

struct ConvArgs {
    const void* src[4];
    int src_c[4];
    int src_stride[4];
    int nsrc;
    const void* weight;
    const float* bias;
    void* out;
    int out_stride;
    int N, H, W, KH, KW, Cin, Cout;
    int stride, Ho, Wo;
    int act, epi;
    const void* aux0;
    const void* aux1;
    int aux0_stride, aux1_stride;
    float out_scale;
    int shuffle2;
};

template <typename T, int BM_, int BN_, int WGM_>
struct ConvCfg {
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = 4 / WGM_;
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int PPR = 8;                        // 16-byte pieces per K-tile row
    static constexpr int BK = PPR * VEC;                 // 64 fp16 / 32 fp32
    static constexpr int RS = BK + VEC;                  // LDS row stride (elements)
    static constexpr int KSTEPS = BK / 16;               // k16 fragments per tile
    static constexpr int WM = BM / WGM, WN = BN / WGN;   // per-wave tile
    static constexpr int MT = WM / 32, NTL = WN / 32;
    static constexpr int A_IT = BM / 32, B_IT = (BN + 31) / 32;   // rows per thread (row = tid/8 + 32*it)
    static constexpr int CRS = BN + VEC;                 // staging row stride
    static constexpr size_t TILE_BYTES = (size_t)2 * (BM + BN) * RS * sizeof(T);
    static constexpr size_t STAGE_BYTES = (size_t)BM * CRS * sizeof(T);
    static constexpr size_t LDS_BYTES = TILE_BYTES > STAGE_BYTES ? TILE_BYTES : STAGE_BYTES;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be whole 32x32 MFMA tiles");
};

__device__ __forceinline__ float activate(float x, int act) {
    switch (act) {
        case S2M2_ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f));
        case S2M2_ACT_RELU: return fmaxf(x, 0.f);
        case S2M2_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
        case S2M2_ACT_TANH: return tanhf(x);
        default: return x;
    }
}

template <typename T> __device__ __forceinline__ Vec16<T> zero_vec() {
    Vec16<T> z;
#pragma unroll
    for (int e = 0; e < (int)(16 / sizeof(T)); ++e) z.v[e] = from_f32<T>(0.f);
    return z;
}

template <typename CFG, typename T>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs p) {
    constexpr int BM = CFG::BM, BN = CFG::BN, VEC = CFG::VEC, RS = CFG::RS, BK = CFG::BK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);                          // [2][BM][RS]
    T* Bs = As + (size_t)2 * BM * RS;                            // [2][BN][RS]
    T* Cs = reinterpret_cast<T*>(smem);                          // [BM][CRS]   (after the K loop)

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = wv % CFG::WGM, wn = wv / CFG::WGM;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int Ktot = p.KH * p.KW * p.Cin;
    const int nkt = (Ktot + BK - 1) / BK;
    const int ph = p.KH / 2, pw = p.KW / 2;

    // ---- loader state: this thread moves piece column `pc` of rows lrow + 32*it
    const int pc = tid & 7, lrow = tid >> 3;
    int ay[CFG::A_IT], ax[CFG::A_IT];
    long long apix[CFG::A_IT];                                   // input pixel (n*H + y)*W + x of the window centre, or -1 past the end
#pragma unroll
    for (int it = 0; it < CFG::A_IT; ++it) {
        const long long m = m0 + lrow + 32 * it;
        if (m < M) {
            const int xo = (int)(m % p.Wo);
            const long long t = m / p.Wo;
            ay[it] = (int)(t % p.Ho) * p.stride;
            ax[it] = xo * p.stride;
            apix[it] = ((t / p.Ho) * p.H + ay[it]) * p.W + ax[it];
        } else { ay[it] = 0; ax[it] = 0; apix[it] = -1; }
    }
    // K position of this thread's piece: element k = kt*BK + pc*VEC  ->  (tap, channel); advanced incrementally
    int kc = pc * VEC, ky = 0, kx = 0;                            // channel within the tap, tap coordinates
    while (kc >= p.Cin) { kc -= p.Cin; if (++kx == p.KW) { kx = 0; ++ky; } }

    Vec16<T> ra[CFG::A_IT], rb[CFG::B_IT];
    auto fetch = [&](int kt) {
        const int k = kt * BK + pc * VEC;
        const bool kvalid = k < Ktot;
        // source of channel kc
        int s = 0, c = kc;
        if (p.nsrc > 1 && c >= p.src_c[0]) { c -= p.src_c[0]; s = 1;
            if (p.nsrc > 2 && c >= p.src_c[1]) { c -= p.src_c[1]; s = 2;
                if (p.nsrc > 3 && c >= p.src_c[2]) { c -= p.src_c[2]; s = 3; } } }
        const T* sp = static_cast<const T*>(p.src[s]);
        const int ss = p.src_stride[s];
        const int dy = ky - ph, dx = kx - pw;
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const int yy = ay[it] + dy, xx = ax[it] + dx;
            const bool ok = kvalid && apix[it] >= 0 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            if (ok) ra[it] = *reinterpret_cast<const Vec16<T>*>(sp + (apix[it] + (long long)dy * p.W + dx) * ss + c);
            else ra[it] = zero_vec<T>();
        }
        const T* wp = static_cast<const T*>(p.weight);
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) {
            const int r = lrow + 32 * it;
            const int co = n0 + r;
            if (r < BN && kvalid && co < p.Cout) rb[it] = *reinterpret_cast<const Vec16<T>*>(wp + (size_t)co * Ktot + k);
            else rb[it] = zero_vec<T>();
        }
        // advance to the next K tile
        kc += BK;
        while (kc >= p.Cin) { kc -= p.Cin; if (++kx == p.KW) { kx = 0; ++ky; } }
    };
    auto stash = [&](int buf) {
        T* a = As + (size_t)buf * BM * RS;
        T* b = Bs + (size_t)buf * BN * RS;
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it)
            *reinterpret_cast<Vec16<T>*>(a + (size_t)(lrow + 32 * it) * RS + pc * VEC) = ra[it];
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) {
            const int r = lrow + 32 * it;
            if (r < BN) *reinterpret_cast<Vec16<T>*>(b + (size_t)r * RS + pc * VEC) = rb[it];
        }
    };

    float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    fetch(0);
    stash(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) fetch(kt + 1);                          // global loads in flight under the MFMAs
        const T* a = As + (size_t)buf * BM * RS + (size_t)(wm * CFG::WM + (lane & 31)) * RS + (lane >> 5) * 8;
        const T* b = Bs + (size_t)buf * BN * RS + (size_t)(wn * CFG::WN + (lane & 31)) * RS + (lane >> 5) * 8;
#pragma unroll
        for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
            Frag<T> xf[CFG::MT], wf[CFG::NTL];
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], a + (size_t)i * 32 * RS + kk * 16);
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j) load_frag(wf[j], b + (size_t)j * 32 * RS + kk * 16);
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                for (int j = 0; j < CFG::NTL; ++j) mma32(acc[i][j], wf[j], xf[i]);    // D[cout][pixel]
        }
        if (kt + 1 < nkt) stash(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue 1: bias, activation, scale in registers -> staging tile Cs[pixel][cout]
    const int hi = lane >> 5;
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i) {
        T* crow = Cs + (size_t)(wm * CFG::WM + i * 32 + (lane & 31)) * CFG::CRS;
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = wn * CFG::WN + j * 32 + 8 * g + 4 * hi;         // local cout of the quad
                const int co = n0 + cl;
                float4_t bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias && co < p.Cout) bv = *reinterpret_cast<const float4_t*>(p.bias + co);
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = activate(acc[i][j][4 * g + e] + bv[e], p.act) * p.out_scale;
                if constexpr (sizeof(T) == 2) {
                    half4_t h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                    *reinterpret_cast<half4_t*>(crow + cl) = h;
                } else {
                    float4_t f = {v[0], v[1], v[2], v[3]};
                    *reinterpret_cast<float4_t*>(crow + cl) = f;
                }
            }
        }
    }
    __syncthreads();

    // ---- epilogue 2: whole-row 16-byte pieces: aux combine, coalesced store
    constexpr int PCR = BN / VEC;                                 // pieces per staged row
    constexpr int TOTAL = BM * PCR;
    T* outp = static_cast<T*>(p.out);
#pragma unroll 2
    for (int q = tid; q < TOTAL; q += 256) {
        const int r = q / PCR, pcc = q - r * PCR;
        const long long m = m0 + r;
        const int co = n0 + pcc * VEC;
        if (m >= M || co >= p.Cout) continue;
        Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(Cs + (size_t)r * CFG::CRS + pcc * VEC);
        if (p.epi != S2M2_EPI_NONE) {
            const Vec16<T> a0 = *reinterpret_cast<const Vec16<T>*>(static_cast<const T*>(p.aux0) + m * p.aux0_stride + co);
            Vec16<T> a1 = a0;
            if (p.epi == S2M2_EPI_GRU || p.epi == S2M2_EPI_GATEMIX)
                a1 = *reinterpret_cast<const Vec16<T>*>(static_cast<const T*>(p.aux1) + m * p.aux1_stride + co);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float x = to_f32(v.v[e]), u = to_f32(a0.v[e]), w = to_f32(a1.v[e]);
                float o;
                if (p.epi == S2M2_EPI_ADD) o = x + u;
                else if (p.epi == S2M2_EPI_MUL) o = x * u;
                else if (p.epi == S2M2_EPI_GRU) o = (1.0f - u) * w + u * x;                      // aux0 = z, aux1 = h, x = q
                else { const float gte = fminf(fmaxf(x, 0.01f), 0.99f); o = gte * u + (1.0f - gte) * w; }   // x = gate
                v.v[e] = from_f32<T>(o);
            }
        }
        long long opix = m;
        int oc = co;
        if (p.shuffle2) {                                         // ConvTranspose2d(k=2, s=2): cout = (dy*2+dx)*C' + c'
            const int sub = co / p.shuffle2;
            oc = co - sub * p.shuffle2;
            const int x = (int)(m % p.Wo);
            const long long t = m / p.Wo;
            const int y = (int)(t % p.Ho);
            const long long n = t / p.Ho;
            opix = (n * (2 * p.Ho) + 2 * y + (sub >> 1)) * (2LL * p.Wo) + 2 * x + (sub & 1);
        }
        *reinterpret_cast<Vec16<T>*>(outp + opix * p.out_stride + oc) = v;
    }
}

template <typename T, int BM, int BN, int WGM>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
    using CFG = ConvCfg<T, BM, BN, WGM>;
    auto kern = conv_igemm_kernel<CFG, T>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)CFG::LDS_BYTES) != hipSuccess)
            return set_error("conv2d: cannot reserve %zu bytes of LDS", CFG::LDS_BYTES);
        attr_done = true;
    }
    const long long M = (long long)a.N * a.Ho * a.Wo;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((a.Cout + BN - 1) / BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), CFG::LDS_BYTES, st, a);
    return check_launch("conv2d");
}

template <typename T>
static int dispatch_conv(const ConvArgs& a, int tile, hipStream_t st) {
    const long long M = (long long)a.N * a.Ho * a.Wo;
    if (tile == 0) {                                              // heuristic: narrow N, else fill the chip
        if (a.Cout <= 32) tile = 3;
        else if (a.Cout <= 64) tile = 4;
        else tile = ((M + 127) / 128) * ((a.Cout + 127) / 128) >= 384 ? 1 : 2;
    }
    switch (tile) {
        case 1: return launch_conv<T, 128, 128, 2>(a, st);
        case 2: return launch_conv<T, 64, 64, 2>(a, st);
        case 3: return launch_conv<T, 128, 32, 4>(a, st);
        case 4: return launch_conv<T, 128, 64, 2>(a, st);
        default: return set_error("conv2d: unknown tile id %d", tile);
    }
}

}  // namespace s2m2

extern "C" int s2m2_conv2d(const s2m2_conv_desc* d, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(d, "conv2d: null descriptor");
    S2M2_REQUIRE(d->nsrc >= 1 && d->nsrc <= 4, "conv2d: nsrc=%d (1..4)", d->nsrc);
    S2M2_REQUIRE(d->weight && d->out, "conv2d: null weight/out");
    S2M2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, "conv2d: bad shape N=%d H=%d W=%d", d->N, d->H, d->W);
    S2M2_REQUIRE((d->KH & 1) && (d->KW & 1) && d->KH <= 7 && d->KW <= 7, "conv2d: kernel %dx%d must be odd and <= 7", d->KH, d->KW);
    S2M2_REQUIRE(d->Cout > 0 && d->Cout % 8 == 0, "conv2d: Cout=%d must be a positive multiple of 8", d->Cout);
    S2M2_REQUIRE(d->out_stride % 8 == 0, "conv2d: out_stride=%d must be a multiple of 8", d->out_stride);
    ConvArgs a;
    a.Cin = 0;
    for (int s = 0; s < 4; ++s) {
        a.src[s] = s < d->nsrc ? d->src[s] : nullptr;
        a.src_c[s] = s < d->nsrc ? d->src_c[s] : 0;
        a.src_stride[s] = s < d->nsrc ? d->src_stride[s] : 0;
        if (s < d->nsrc) {
            S2M2_REQUIRE(d->src[s], "conv2d: src[%d] is null", s);
            S2M2_REQUIRE(d->src_c[s] > 0 && d->src_c[s] % 8 == 0 && d->src_stride[s] % 8 == 0 && d->src_stride[s] >= d->src_c[s],
                         "conv2d: src[%d] channels=%d stride=%d must be multiples of 8", s, d->src_c[s], d->src_stride[s]);
            a.Cin += d->src_c[s];
        }
    }
    S2M2_REQUIRE(d->act >= S2M2_ACT_NONE && d->act <= S2M2_ACT_TANH, "conv2d: unknown activation %d", d->act);
    S2M2_REQUIRE(d->epi >= S2M2_EPI_NONE && d->epi <= S2M2_EPI_GATEMIX, "conv2d: unknown epilogue %d", d->epi);
    if (d->epi != S2M2_EPI_NONE) {
        S2M2_REQUIRE(d->aux0 && d->aux0_stride % 8 == 0, "conv2d: epilogue %d needs aux0 (stride multiple of 8)", d->epi);
        if (d->epi == S2M2_EPI_GRU || d->epi == S2M2_EPI_GATEMIX)
            S2M2_REQUIRE(d->aux1 && d->aux1_stride % 8 == 0, "conv2d: epilogue %d needs aux1 (stride multiple of 8)", d->epi);
        S2M2_REQUIRE(!d->shuffle2, "conv2d: aux epilogues are not supported with shuffle2");
    }
    if (d->shuffle2)
        S2M2_REQUIRE(d->shuffle2 % 8 == 0 && d->Cout == 4 * d->shuffle2 && d->KH == 1 && d->KW == 1 && d->stride == 1,
                     "conv2d: shuffle2=%d needs a 1x1 stride-1 kernel and Cout == 4*shuffle2 (multiple of 8)", d->shuffle2);
    S2M2_REQUIRE(d->stride == 1 || d->stride == 2, "conv2d: stride=%d (1 or 2)", d->stride);
    a.nsrc = d->nsrc; a.weight = d->weight; a.bias = d->bias; a.out = d->out; a.out_stride = d->out_stride;
    a.N = d->N; a.H = d->H; a.W = d->W; a.KH = d->KH; a.KW = d->KW; a.Cout = d->Cout;
    a.act = d->act; a.epi = d->epi; a.aux0 = d->aux0; a.aux1 = d->aux1;
    a.aux0_stride = d->aux0_stride; a.aux1_stride = d->aux1_stride;
    a.out_scale = d->out_scale; a.shuffle2 = d->shuffle2;
    a.stride = d->stride; a.Ho = (d->H + d->stride - 1) / d->stride; a.Wo = (d->W + d->stride - 1) / d->stride;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->dtype == S2M2_F16) return dispatch_conv<half_t>(a, d->tile, st);
    if (d->dtype == S2M2_F32) return dispatch_conv<float>(a, d->tile, st);
    return set_error("conv2d: unsupported dtype %d", d->dtype);
}
