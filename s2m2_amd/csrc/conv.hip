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
#include "plan.h"
#include <stdlib.h>

// compile-time ablation switches (tools/conv_ablate.py; never set in the shipped library):
// 2 no MFMA, 8 no fragment reads, 16 no bias/activation math, 32 no global stores, 64 no K loop at all
// (1 = no global loads / 4 = no LDS stash inside the K loop were retired with the uniform fetch/stash loop)
#ifndef S2M2_CONV_DBG
#define S2M2_CONV_DBG 0
#endif
#include "epilogue.h"

// which K loops use untracked loads + counted waits (common.h): bit 0 the halo kernel (ring variants), bit 1 the v1 kernel
#ifndef S2M2_ASYNC_LOADS
#define S2M2_ASYNC_LOADS 0      // measured neutral end to end on MI355X (same-box A/B, profiles/r01/async_loads_ab.txt): tracked loads stay the default
#endif

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
    int korder;                 // 0: K = (ky, kx, c);  1: K = (ky, c / CH, kx, c % CH), CH = 64 bytes of channels (L1 reuse along kx)
    const void* zero;           // 256 zero bytes in global memory (what out-of-range pieces read)
    const float* ln_wsum;       // non-null: pre-LayerNorm (no affine) of the input rows folded into the GEMM, see LnStats
    float ln_eps;
    int ksplit;                 // S2M2_EPI_DUALMIX: K index where the second GEMM (second accumulator, bias2) starts
    const float* bias2;
    int pool2;                  // 1x1 layer behind AvgPool2d(2): a row is the mean of 4 input pixels (stride = 2, Ho = H / 2, Wo = W / 2)
    int epi_cout0;              // v5 only: the epilogue applies to cout blocks at or above this cout (0: all)
};

template <typename T, int BM_, int BN_, int WGM_, int PPR_ = 8, int NPF_ = 1, int NWAVES_ = 4>
struct ConvCfg {
    static constexpr int NT = 64 * NWAVES_;              // threads per block
    static constexpr int NPF = NPF_;                     // K tiles requested ahead, in registers (short-K layers: the whole K at once)
    static constexpr int BM = BM_, BN = BN_, WGM = WGM_, WGN = NWAVES_ / WGM_;
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int PPR = PPR_;                     // 16-byte pieces per K-tile row (8: 128-byte rows, 4: 64-byte rows)
    static constexpr int RPI = NT / PPR;                 // tile rows covered by one pass of the loader threads
    static constexpr int BK = PPR * VEC;                 // 64 fp16 / 32 fp32
    static constexpr int RS = BK + VEC;                  // LDS row stride (elements)
    static constexpr int KSTEPS = BK / 16;               // k16 fragments per tile
    static constexpr int WM = BM / WGM, WN = BN / WGN;   // per-wave tile
    static constexpr int MT = WM / 32, NTL = WN / 32;
    static constexpr int A_IT = (BM + RPI - 1) / RPI, B_IT = (BN + RPI - 1) / RPI;   // rows per thread (row = tid/PPR + RPI*it)
    static constexpr int CRS = BN + VEC;                 // staging row stride
    static constexpr size_t TILE_BYTES = (size_t)2 * (BM + BN) * RS * sizeof(T);
    static constexpr size_t STAGE_BYTES = (size_t)BM * CRS * sizeof(T);
    static constexpr size_t LDS_BYTES = TILE_BYTES > STAGE_BYTES ? TILE_BYTES : STAGE_BYTES;
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be whole 32x32 MFMA tiles");
};

template <typename T> __device__ __forceinline__ Vec16<T> zero_vec() {
    Vec16<T> z;
#pragma unroll
    for (int e = 0; e < (int)(16 / sizeof(T)); ++e) z.v[e] = from_f32<T>(0.f);
    return z;
}

// Aux tensors of the epilogue (residual / gate operands), one 16-byte piece per staged piece of this thread: requested right
// after the K loop, so their latency runs under the activation math, the LDS staging and its barrier instead of in front of
// every store.  Kept in registers only while that is cheap (at most 4 pieces per thread); larger tiles load inside the loop.
template <typename CFG, typename T, int MAXNP = 4>
struct AuxRegs {
    static constexpr int PCR = CFG::BN / CFG::VEC;                // pieces per staged row
    static constexpr int TOTAL = CFG::BM * PCR;
    static constexpr int NP = (TOTAL + CFG::NT - 1) / CFG::NT;    // pieces per thread
    static constexpr bool ON = NP <= MAXNP;
    raw16_t a0[ON ? NP : 1], a1[ON ? NP : 1];

    // pix(r, m): global output pixel of staged row r (false: outside the image / past M)
    // ONE: the caller launches with one-operand epilogues only (a1 stays a constant zero: no registers)
    template <bool ONE = false, typename PIX>
    __device__ __forceinline__ void prefetch(const ConvArgs& p, int tid, int n0, PIX pix, int epi_override = -1) {
        if constexpr (ON) {
            const int epi = epi_override >= 0 ? epi_override : p.epi;
            if (epi == S2M2_EPI_NONE) return;
            const bool two = !ONE && (epi == S2M2_EPI_GRU || epi == S2M2_EPI_GATEMIX || epi == S2M2_EPI_DUALMIX);
#pragma unroll
            for (int it = 0; it < NP; ++it) {
                const int q = tid + CFG::NT * it, r = q / PCR, pcc = q - r * PCR;
                const int co = n0 + pcc * CFG::VEC;
                long long m;
                const bool ok = q < TOTAL && pix(r, m) && co < p.Cout;
                a0[it] = global_load16(ok ? static_cast<const T*>(p.aux0) + m * p.aux0_stride + co : static_cast<const T*>(p.zero));
                a1[it] = raw16_t{0.f, 0.f, 0.f, 0.f};             // (copying a0 here would wait for its load)
                if (two) a1[it] = global_load16(ok ? static_cast<const T*>(p.aux1) + m * p.aux1_stride + co : static_cast<const T*>(p.zero));
            }
        }
    }
};

// epilogue 2: the staged tile comes back as 16-byte pieces of whole pixel rows: aux combine, coalesced store
template <typename CFG, typename T, typename AX, typename PIX>
__device__ __forceinline__ void store_tile(const ConvArgs& p, const T* Cs, int tid, int n0, const AX& aux, PIX pix, int epi_override = -1) {
    constexpr int VEC = CFG::VEC, PCR = AX::PCR;
    const int epi = epi_override >= 0 ? epi_override : p.epi;
    T* outp = static_cast<T*>(p.out);
#pragma unroll
    for (int it = 0; it < AX::NP; ++it) {
        const int q = tid + CFG::NT * it, r = q / PCR, pcc = q - r * PCR;
        const int co = n0 + pcc * VEC;
        long long m;
        if (q >= AX::TOTAL || !pix(r, m) || co >= p.Cout) continue;
        Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(Cs + (size_t)r * CFG::CRS + pcc * VEC);
        if (epi != S2M2_EPI_NONE) {
            Vec16<T> a0, a1;
            if constexpr (AX::ON) {
                a0 = __builtin_bit_cast(Vec16<T>, aux.a0[it]);
                a1 = __builtin_bit_cast(Vec16<T>, aux.a1[it]);
            } else {
                a0 = *reinterpret_cast<const Vec16<T>*>(static_cast<const T*>(p.aux0) + m * p.aux0_stride + co);
                a1 = a0;
                if (epi == S2M2_EPI_GRU || epi == S2M2_EPI_GATEMIX)
                    a1 = *reinterpret_cast<const Vec16<T>*>(static_cast<const T*>(p.aux1) + m * p.aux1_stride + co);
            }
            aux_combine(v, epi, a0, a1);
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
        if (!(S2M2_CONV_DBG & 32)) *reinterpret_cast<Vec16<T>*>(outp + opix * p.out_stride + oc) = v;
    }
}

// staged row -> output pixel maps: linear kernels (row r of the tile is pixel m0 + r) ...
struct LinearPix {
    long long m0, M;
    __device__ __forceinline__ bool operator()(int r, long long& m) const { m = m0 + r; return m < M; }
};
// ... and the halo kernel (row r = patch row * 32 + column)
template <int PW = 32>
struct PatchPixT {
    int n, y0, x0, H, W;
    __device__ __forceinline__ bool operator()(int r, long long& m) const {
        const int py = r / PW;                                     // (constant divisor: a shift for the 32-wide patches)
        const int yy = y0 + py, xx = x0 + (r - py * PW);
        m = ((long long)n * H + yy) * W + xx;
        return yy < H && xx < W;
    }
};
using PatchPix = PatchPixT<32>;

// the pre-LN layers of the model are the QKV projections (no activation) and the first FFN layer (GELU)
template <typename CFG, typename T>
__device__ __forceinline__ void stage_tile_ln(const ConvArgs& p, const float16_t (&acc)[CFG::MT][CFG::NTL], T* Cs, const CoutRegs<CFG>& bias,
                                              int wm, int wn, int lane, const LnRow* ln, const CoutRegs<CFG>& wsum) {
    if (p.act == S2M2_ACT_GELU) stage_tile<CFG, T, S2M2_ACT_GELU, true>(acc, Cs, bias, p.out_scale, wm, wn, lane, ln, &wsum);
    else stage_tile<CFG, T, S2M2_ACT_NONE, true>(acc, Cs, bias, p.out_scale, wm, wn, lane, ln, &wsum);
}

template <typename CFG, typename T>
__device__ __forceinline__ void stage_tile_act(const ConvArgs& p, const float16_t (&acc)[CFG::MT][CFG::NTL], T* Cs, const CoutRegs<CFG>& bias,
                                               int wm, int wn, int lane) {
    switch (p.act) {                                              // block-uniform
        case S2M2_ACT_GELU: stage_tile<CFG, T, S2M2_ACT_GELU>(acc, Cs, bias, p.out_scale, wm, wn, lane); break;
        case S2M2_ACT_RELU: stage_tile<CFG, T, S2M2_ACT_RELU>(acc, Cs, bias, p.out_scale, wm, wn, lane); break;
        case S2M2_ACT_SIGMOID: stage_tile<CFG, T, S2M2_ACT_SIGMOID>(acc, Cs, bias, p.out_scale, wm, wn, lane); break;
        case S2M2_ACT_TANH: stage_tile<CFG, T, S2M2_ACT_TANH>(acc, Cs, bias, p.out_scale, wm, wn, lane); break;
        default: stage_tile<CFG, T, S2M2_ACT_NONE>(acc, Cs, bias, p.out_scale, wm, wn, lane); break;
    }
}

// K cursor of one lane: which (tap, channel) its 16-byte piece of the NEXT tile comes from.  Two K orders (weights are packed to
// match, the weight side only ever sees the linear K index):
//   0  (ky, kx, c)                  any Cin
//   1  (ky, c / CH, kx, c % CH)     CH = 64 bytes of channels, Cin % CH == 0: consecutive tiles sweep kx over the SAME cache lines
//      (pixel x+1 at tap kx is pixel x at tap kx+1), so the three horizontal taps of a 3x3 kernel hit in L1 instead of L2.
template <int CH>
struct KCursor {
    int ky, kx, kc;
    __device__ __forceinline__ void init(const ConvArgs& p, int elem) {          // elem: K offset of the piece inside tile 0
        ky = 0; kx = 0;
        if (p.korder) { kc = elem % CH; step(p, elem / CH); }
        else { kc = elem; norm(p); }
    }
    __device__ __forceinline__ void norm(const ConvArgs& p) {
        while (kc >= p.Cin) { kc -= p.Cin; if (++kx == p.KW) { kx = 0; ++ky; } }
    }
    __device__ __forceinline__ void step(const ConvArgs& p, int n) {
        for (int i = 0; i < n; ++i) {
            if (++kx == p.KW) { kx = 0; kc += CH; if (kc >= p.Cin) { kc -= p.Cin; ++ky; } }
        }
    }
    template <int BK> __device__ __forceinline__ void advance(const ConvArgs& p) {
        if (p.korder) step(p, BK / CH);
        else { kc += BK; norm(p); }
    }
};

// Per-thread state of the global -> register -> LDS tile loader: piece column `pc` of rows lrow + RPI*it.
//  * everything that does not change along K is computed once (pixel index, a bit mask of the taps that fall inside the image,
//    weight row pointers); a fetch is ~7 VALU ops per 16-byte piece: 24-bit multiply-add for the element offset, one bit test,
//    one 64-bit select between the real address and a 16-byte zero block (zero padding, K / Cout / M tails) -- no branches;
//  * kernel-argument arrays are never indexed with a runtime value (that becomes memory loads + vmcnt(0) in front of every tile):
//    the per-lane source is picked with masked telescoping sums over scalars.
template <typename CFG, typename T, bool POOL = false>
struct ConvLoader {
    static constexpr bool V1A = (S2M2_ASYNC_LOADS & 2) != 0;
    static constexpr int NQ = POOL ? 4 : 1;    // POOL (AvgPool2d(2) folded into a 1x1 layer): a piece is the mean of the pieces of 4 pixels
    static constexpr int VEC = CFG::VEC, BK = CFG::BK, RS = CFG::RS, BN = CFG::BN;
    int pc, lrow, Ktot;
    int apix[CFG::A_IT];                       // input pixel (n*H + y)*W + x of the window centre
    unsigned tapmask[CFG::A_IT];               // bit (ky*KW + kx): that tap of this row is inside the image (0 for rows past M)
    const T* wrow[CFG::B_IT];                  // weight row of this thread (nullptr: row past Cout / BN)
    KCursor<4 * VEC> cur;                      // K position of this thread's piece: channel within the tap, tap coordinates
    const T *s0, *s1, *s2, *s3;                // sources / strides / cumulative channel counts as named scalars
    int st0, st1, st2, st3, c0n, c1n, c2n;
    raw16_t ra[CFG::NPF][CFG::A_IT * NQ], rb[CFG::NPF][CFG::B_IT];

    __device__ __forceinline__ void init(const ConvArgs& p, int tid, long long m0, int n0, long long M, int Ktot_) {
        pc = tid % CFG::PPR; lrow = tid / CFG::PPR; Ktot = Ktot_;
        s0 = static_cast<const T*>(p.src[0]); s1 = static_cast<const T*>(p.src[1]);
        s2 = static_cast<const T*>(p.src[2]); s3 = static_cast<const T*>(p.src[3]);
        st0 = p.src_stride[0]; st1 = p.src_stride[1]; st2 = p.src_stride[2]; st3 = p.src_stride[3];
        c0n = p.src_c[0]; c1n = c0n + p.src_c[1]; c2n = c1n + p.src_c[2];
        const int ph = p.KH / 2, pw = p.KW / 2;
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const long long m = m0 + lrow + CFG::RPI * it;
            apix[it] = 0; tapmask[it] = 0u;
            if (m < M && lrow + CFG::RPI * it < CFG::BM) {                                          // M < 2^24 (checked on the host): 32-bit divisions
                const unsigned mu = (unsigned)m;
                const unsigned t = mu / (unsigned)p.Wo;
                const int xo = (int)(mu - t * (unsigned)p.Wo);
                const unsigned n = t / (unsigned)p.Ho;
                const int y = (int)(t - n * (unsigned)p.Ho) * p.stride, x = xo * p.stride;
                apix[it] = ((int)n * p.H + y) * p.W + x;
                unsigned mk = 0u;
                for (int a = 0; a < p.KH; ++a)
                    for (int b = 0; b < p.KW; ++b) {
                        const int yy = y + a - ph, xx = x + b - pw;
                        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mk |= 1u << (a * p.KW + b);
                    }
                tapmask[it] = mk;
            }
        }
        const T* wp = static_cast<const T*>(p.weight);
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) {
            const int r = lrow + CFG::RPI * it;
            const int co = n0 + r;
            wrow[it] = (r < BN && co < p.Cout) ? wp + (size_t)co * Ktot + pc * VEC : nullptr;
        }
        cur.init(p, pc * VEC);
    }

    // issue the loads of K tile kt (element k = kt*BK + pc*VEC -> (tap, channel)), then advance to the next tile
    __device__ __forceinline__ void fetch(const ConvArgs& p, int kt, int slot) {
        const T* zp = static_cast<const T*>(p.zero);
        const bool kvalid = kt * BK + pc * VEC < Ktot;
        const int kc = cur.kc, ky = cur.ky, kx = cur.kx;
        const T* sp = s0;
        unsigned ss = (unsigned)st0, c = (unsigned)kc;
        if (p.nsrc > 1) {                                         // wave-uniform branch on a scalar
            const bool g0 = kc >= c0n, g1 = kc >= c1n, g2 = kc >= c2n;
            const long long d1 = (const char*)s1 - (const char*)s0, d2 = (const char*)s2 - (const char*)s1, d3 = (const char*)s3 - (const char*)s2;
            sp = reinterpret_cast<const T*>((const char*)s0 + ((g0 ? d1 : 0) + (g1 ? d2 : 0) + (g2 ? d3 : 0)));
            ss = (unsigned)(st0 + (g0 ? st1 - st0 : 0) + (g1 ? st2 - st1 : 0) + (g2 ? st3 - st2 : 0));
            c = (unsigned)(kc - ((g0 ? c0n : 0) + (g1 ? c1n - c0n : 0) + (g2 ? c2n - c1n : 0)));
        }
        const int tapoff = (ky - p.KH / 2) * p.W + (kx - p.KW / 2);
        const unsigned tapbit = kvalid ? 1u << (ky * p.KW + kx) : 0u;
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const unsigned e = __umul24((unsigned)(apix[it] + tapoff), ss) + c;      // pixel < 2^24, stride < 2^24, numel < 2^31
            const bool ok = (tapmask[it] & tapbit) != 0;
            const T* src = ok ? sp + e : zp;
            if constexpr (!POOL) {
                global_load16_async<V1A>(ra[slot][it], src);
            } else {                                              // pixels (2y, 2x), (2y, 2x+1), (2y+1, 2x), (2y+1, 2x+1): all inside (H, W >= 2*Ho, 2*Wo)
                const unsigned dx = ok ? ss : 0u, dy = ok ? (unsigned)p.W * ss : 0u;
                global_load16_async<V1A>(ra[slot][4 * it + 0], src);
                global_load16_async<V1A>(ra[slot][4 * it + 1], src + dx);
                global_load16_async<V1A>(ra[slot][4 * it + 2], src + dy);
                global_load16_async<V1A>(ra[slot][4 * it + 3], src + dy + dx);
            }
        }
        const int koff = kt * BK;
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) {
            const T* src = (kvalid && wrow[it]) ? wrow[it] + koff : zp;
            global_load16_async<V1A>(rb[slot][it], src);
        }
        cur.template advance<BK>(p);
    }

    // the loads of a K tile are untracked (common.h: global_load16_async): every fetch issues exactly A_IT + B_IT of them (tiles past
    // the end of K read the zero page), so "the tile in `slot` has landed" = at most (NPF-1) younger tiles outstanding
    static constexpr int LOADS_PER_TILE = CFG::A_IT * NQ + CFG::B_IT;
    __device__ __forceinline__ void stash(T* a, T* b, int slot) {
        wait_vmcnt<(CFG::NPF - 1) * LOADS_PER_TILE, V1A>();
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const int r = lrow + CFG::RPI * it;
            if constexpr (!POOL) {
                settle(ra[slot][it]);
                if (r < CFG::BM) *reinterpret_cast<raw16_t*>(a + (size_t)r * RS + pc * VEC) = ra[slot][it];
            } else {                                              // the mean exactly as K7's AvgPool kernel forms and rounds it (upsample.hip)
#pragma unroll
                for (int q = 0; q < 4; ++q) settle(ra[slot][4 * it + q]);
                const Vec16<T> va = __builtin_bit_cast(Vec16<T>, ra[slot][4 * it + 0]), vb = __builtin_bit_cast(Vec16<T>, ra[slot][4 * it + 1]);
                const Vec16<T> vc = __builtin_bit_cast(Vec16<T>, ra[slot][4 * it + 2]), vd = __builtin_bit_cast(Vec16<T>, ra[slot][4 * it + 3]);
                Vec16<T> o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>((to_f32(va.v[e]) + to_f32(vb.v[e]) + to_f32(vc.v[e]) + to_f32(vd.v[e])) * 0.25f);
                if (r < CFG::BM) *reinterpret_cast<Vec16<T>*>(a + (size_t)r * RS + pc * VEC) = o;
            }
        }
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) {
            settle(rb[slot][it]);
            const int r = lrow + CFG::RPI * it;
            if (r < BN) *reinterpret_cast<raw16_t*>(b + (size_t)r * RS + pc * VEC) = rb[slot][it];
        }
    }
    __device__ __forceinline__ void drain() {                     // tail requests (zero page): land before their registers are reused
        wait_vmcnt<0, V1A>();
#pragma unroll
        for (int f = 0; f < CFG::NPF; ++f) {
#pragma unroll
            for (int it = 0; it < CFG::A_IT * NQ; ++it) settle(ra[f][it]);
#pragma unroll
            for (int it = 0; it < CFG::B_IT; ++it) settle(rb[f][it]);
        }
    }
};

// MODE 0: plain; 1: pre-LayerNorm folded in (ln_wsum); 2: two GEMMs over consecutive K ranges of the same rows into two accumulators,
// combined by the S2M2_EPI_DUALMIX epilogue (the gate and fusion heads of FeatureFusion in one launch); 3: AvgPool2d(2) folded into the
// A-operand load of a 1x1 layer (pool2: the down_conv of Unet / MRT, unet.py / stacked_MRT.py `nn.AvgPool2d(2), nn.Conv2d(., ., 1)`)
template <typename CFG, typename T, int MODE = 0>
__global__ __launch_bounds__(CFG::NT) void conv_igemm_kernel(ConvArgs p) {
    constexpr bool LN = MODE == 1, DUAL = MODE == 2;
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

    // ---- loader: this thread moves piece column `pc` of rows lrow + 32*it (see ConvLoader)
    ConvLoader<CFG, T, MODE == 3> ld;
    ld.init(p, tid, m0, n0, M, Ktot);

    float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float16_t acc2[DUAL ? CFG::MT : 1][DUAL ? CFG::NTL : 1];     // second GEMM (K >= ksplit)
    if constexpr (DUAL) {
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    }
    float ln_s[CFG::MT], ln_q[CFG::MT], ln_shift[CFG::MT];
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i) ln_s[i] = ln_q[i] = ln_shift[i] = 0.f;
    CoutRegs<CFG> bias, wsum;                                     // requested now, used after the K loop (wsum: rowsum(W) or the second bias)
    if constexpr (!DUAL) bias.load(p.bias, p.zero, p.Cout, n0, wn, lane);
    if constexpr (LN) wsum.load(p.ln_wsum, p.zero, p.Cout, n0, wn, lane);

    // K tiles are requested NPF ahead into a ring of register slots (slot = tile % NPF, static after unrolling): for the short-K
    // layers (1x1, K <= NPF tiles) the whole K of the block is in flight at once -- one memory latency per block instead of one per tile
    constexpr int NPF = CFG::NPF;
#pragma unroll
    for (int f = 0; f < NPF; ++f) ld.fetch(p, f, f);
    ld.stash(As, Bs, 0);
    __syncthreads();
    if constexpr (LN && sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i) ln_shift[i] = to_f32(As[(size_t)(wm * CFG::WM + i * 32 + (lane & 31)) * RS]);
    }
    const int nkt_run = (S2M2_CONV_DBG & 64) ? 0 : nkt;
    for (int kt0 = 0; kt0 < nkt_run; kt0 += NPF) {
#pragma unroll
        for (int f = 0; f < NPF; ++f) {
            const int kt = kt0 + f;
            if (kt >= nkt_run) break;
            const int buf = kt & 1;
            ld.fetch(p, kt + NPF, f);                             // slot f was stashed one iteration ago: refill (zero page past the end)
            const T* a = As + (size_t)buf * BM * RS + (size_t)(wm * CFG::WM + (lane & 31)) * RS + (lane >> 5) * 8;
            const T* b = Bs + (size_t)buf * BN * RS + (size_t)(wn * CFG::WN + (lane & 31)) * RS + (lane >> 5) * 8;
#pragma unroll
            for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
                Frag<T> xf[CFG::MT], wf[CFG::NTL];
#pragma unroll
                for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], (S2M2_CONV_DBG & 8) ? a : a + (size_t)i * 32 * RS + kk * 16);
#pragma unroll
                for (int j = 0; j < CFG::NTL; ++j) load_frag(wf[j], (S2M2_CONV_DBG & 8) ? b : b + (size_t)j * 32 * RS + kk * 16);
                if constexpr (LN) {
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i) ln_accumulate(xf[i], ln_s[i], ln_q[i], ln_shift[i]);
                }
                if (DUAL && kt * BK >= p.ksplit) {                  // block-uniform: the K tile belongs to the second GEMM
                    if constexpr (DUAL) {
#pragma unroll
                        for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                            for (int j = 0; j < CFG::NTL; ++j) mma32(acc2[i][j], wf[j], xf[i]);
                    }
                } else if (!(S2M2_CONV_DBG & 2)) {
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                        for (int j = 0; j < CFG::NTL; ++j) mma32(acc[i][j], wf[j], xf[i]);    // D[cout][pixel]
                } else {
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                        for (int j = 0; j < CFG::NTL; ++j) acc[i][j][0] += to_f32(wf[j].v[0]) + to_f32(xf[i].v[0]);
                }
            }
            ld.stash(As + (size_t)(buf ^ 1) * BM * RS, Bs + (size_t)(buf ^ 1) * BN * RS, (f + 1) % NPF);   // (past the end: zeros, idle buffer)
            __syncthreads();
        }
    }
    ld.drain();
    if constexpr (DUAL) {                                         // two accumulator sets: no registers to spare during the K loop
        bias.load(p.bias, p.zero, p.Cout, n0, wn, lane);
        wsum.load(p.bias2, p.zero, p.Cout, n0, wn, lane);
    }

    // ---- epilogue 1: bias, activation, scale in registers -> staging tile Cs[pixel][cout]
    const LinearPix pix{m0, M};
    AuxRegs<CFG, T> aux;
    if constexpr (!DUAL) aux.prefetch(p, tid, n0, pix);           // (the dual-GEMM epilogue has no registers to park them in)
    if constexpr (LN) {
        LnRow ln[CFG::MT];
        const float inv = 1.0f / (float)Ktot;
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i) {
            const float s = ln_s[i] + __shfl_xor(ln_s[i], 32), q = ln_q[i] + __shfl_xor(ln_q[i], 32);
            const float mean = s * inv;                            // relative to ln_shift (0 for fp16 rows)
            ln[i].mean = mean + ln_shift[i];
            ln[i].rstd = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, q * inv), 0.f) + p.ln_eps);
        }
        stage_tile_ln<CFG, T>(p, acc, Cs, bias, wm, wn, lane, ln, wsum);
    } else if constexpr (DUAL) {
        // out = (acc2 + bias2) + mix(clamp(sigmoid(acc + bias)), aux0, aux1): both tiles go through the one staging buffer in turn,
        // the mix of a thread's pieces waits in registers in between (rounded to T where the separate launches stored it)
        using AX = AuxRegs<CFG, T>;
        static_assert(AX::ON, "the dual-GEMM epilogue keeps the aux pieces of a thread in registers");
        stage_tile<CFG, T, S2M2_ACT_SIGMOID>(acc, Cs, bias, 1.0f, wm, wn, lane);
        __syncthreads();
        Vec16<T> mix[AX::NP];
#pragma unroll
        for (int it = 0; it < AX::NP; ++it) {
            const int q = tid + CFG::NT * it, r = q / AX::PCR, pcc = q - r * AX::PCR;
            const int co = n0 + pcc * VEC;
            long long m;
            const bool ok = q < AX::TOTAL && pix(r, m) && co < p.Cout;
            const raw16_t u0 = global_load16(ok ? static_cast<const T*>(p.aux0) + m * p.aux0_stride + co : static_cast<const T*>(p.zero));
            const raw16_t u1 = global_load16(ok ? static_cast<const T*>(p.aux1) + m * p.aux1_stride + co : static_cast<const T*>(p.zero));
            mix[it] = *reinterpret_cast<const Vec16<T>*>(Cs + (size_t)(q < AX::TOTAL ? r : 0) * CFG::CRS + pcc * VEC);
            aux_combine(mix[it], S2M2_EPI_GATEMIX, __builtin_bit_cast(Vec16<T>, u0), __builtin_bit_cast(Vec16<T>, u1));
        }
        __syncthreads();
        stage_tile<CFG, T, S2M2_ACT_NONE>(acc2, Cs, wsum, 1.0f, wm, wn, lane);
        __syncthreads();
        T* outp = static_cast<T*>(p.out);
#pragma unroll
        for (int it = 0; it < AX::NP; ++it) {
            const int q = tid + CFG::NT * it, r = q / AX::PCR, pcc = q - r * AX::PCR;
            const int co = n0 + pcc * VEC;
            long long m;
            if (q >= AX::TOTAL || !pix(r, m) || co >= p.Cout) continue;
            Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(Cs + (size_t)r * CFG::CRS + pcc * VEC);
            aux_combine(v, S2M2_EPI_ADD, mix[it], mix[it]);
            *reinterpret_cast<Vec16<T>*>(outp + m * p.out_stride + co) = v;
        }
        return;
    } else {
        stage_tile_act<CFG, T>(p, acc, Cs, bias, wm, wn, lane);
    }
    __syncthreads();

    store_tile<CFG, T>(p, Cs, tid, n0, aux, pix);
}

// ---------------------------------------------------------------------------------------------------------------
// v2 pipeline: global -> LDS DIRECT (global_load_lds_dwordx4, no VGPR staging, no ds_write) into a ring of NS stages.
//   stage  = KP planes x (BM + BN) rows x 64 bytes: a plane holds 4 16-byte pieces (32 fp16 / 16 fp32 of K) per row, rows dense
//            (an LDS-direct wave instruction fills 1024 contiguous bytes = 16 rows), bank conflicts of the ds_read_b128
//            fragment reads removed by an XOR swizzle applied on the SOURCE side: slot s of row r holds piece s ^ ((r>>2)&3),
//            which is constant per lane, so every lane keeps ONE K cursor per plane;
//   ring   = tile kt+NS-1 is requested while tile kt is multiplied: NS-1 tiles of latency cover, ONE block barrier per tile,
//            s_waitcnt vmcnt(N) counted (never 0 inside the loop); K tails / zero padding read the zero block, so every
//            iteration issues the same number of loads and the counted wait stays valid.
// Everything else (operand roles, epilogues) is shared with the v1 kernel above.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BM_, int BN_, int KP_, int NS_>
struct ConvCfg2 {
    static constexpr int BM = BM_, BN = BN_, KP = KP_, NS = NS_, WGM = 2, WGN = 2, NT = 256;
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int PBK = 4 * VEC;                  // K elements per plane (64 bytes)
    static constexpr int BK = KP * PBK;
    static constexpr int PSTEPS = PBK / 16 > 0 ? PBK / 16 : 1;     // k16 steps per plane: 2 (fp16), 1 (fp32)
    static constexpr int WM = BM / 2, WN = BN / 2, MT = WM / 32, NTL = WN / 32;
    static constexpr int A_INS = BM / 64, B_INS = BN / 64;        // LDS-direct instructions per wave and plane (16 rows each)
    static constexpr int LPS = KP * (A_INS + B_INS);             // loads per lane and stage
    static constexpr int PLANE_BYTES = (BM + BN) * 64;
    static constexpr int STAGE_BYTES = KP * PLANE_BYTES;
    static constexpr int CRS = BN + VEC;
    static constexpr size_t TILE_BYTES = (size_t)NS * STAGE_BYTES;
    static constexpr size_t STAGE_C_BYTES = (size_t)BM * CRS * sizeof(T);
    static constexpr size_t LDS_BYTES = TILE_BYTES > STAGE_C_BYTES ? TILE_BYTES : STAGE_C_BYTES;
};

// LDS fragment reads of the v2 kernel are inline asm on purpose: the compiler orders every ds_read it can see behind ALL pending
// LDS-direct loads (it cannot tell the ring slots apart) with s_waitcnt vmcnt(0), which would drain the ring every iteration.
// Here the protocol is explicit: counted vmcnt + s_barrier before a slot is read, lgkmcnt(0) before the fragments are used.
__device__ __forceinline__ raw16_t lds_read16(unsigned addr) {
    raw16_t v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}
template <typename T> struct RawFrag;                                  // the registers of one k16 fragment as raw 16-byte reads
template <> struct RawFrag<half_t> { raw16_t a; };
template <> struct RawFrag<float> { raw16_t a, b; };
__device__ __forceinline__ void issue_frag(RawFrag<half_t>& f, unsigned rowaddr, int step, int hi, int sw) {
    f.a = lds_read16(rowaddr + (((step * 2 + hi) ^ sw) << 4));
}
__device__ __forceinline__ void issue_frag(RawFrag<float>& f, unsigned rowaddr, int step, int hi, int sw) {
    (void)step;
    f.a = lds_read16(rowaddr + (((2 * hi) ^ sw) << 4));
    f.b = lds_read16(rowaddr + (((2 * hi + 1) ^ sw) << 4));
}
__device__ __forceinline__ void settle(RawFrag<half_t>& f) { asm volatile("" : "+v"(f.a)); }
__device__ __forceinline__ void settle(RawFrag<float>& f) { asm volatile("" : "+v"(f.a), "+v"(f.b)); }
__device__ __forceinline__ Frag<half_t> to_frag(const RawFrag<half_t>& r) { Frag<half_t> f; f.v = __builtin_bit_cast(half8_t, r.a); return f; }
__device__ __forceinline__ Frag<float> to_frag(const RawFrag<float>& r) {
    Frag<float> f;
    f.v[0] = r.a[0]; f.v[1] = r.a[1]; f.v[2] = r.a[2]; f.v[3] = r.a[3];
    f.v[4] = r.b[0]; f.v[5] = r.b[1]; f.v[6] = r.b[2]; f.v[7] = r.b[3];
    return f;
}

template <typename CFG, typename T>
__global__ __launch_bounds__(256) void conv_igemm2_kernel(ConvArgs p) {
    constexpr int BM = CFG::BM, BN = CFG::BN, VEC = CFG::VEC, KP = CFG::KP, NS = CFG::NS;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    T* Cs = reinterpret_cast<T*>(smem);                          // [BM][CRS]   (after the K loop)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & 1, wn = wv >> 1;
    const long long M = (long long)p.N * p.Ho * p.Wo;
    const long long m0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int Ktot = p.KH * p.KW * p.Cin;
    const int nkt = (Ktot + CFG::BK - 1) / CFG::BK;

    // ---- loader state (per lane): piece column q of rows  wv*(rows/4) + 16*i + (lane>>2)
    const int q = (lane & 3) ^ ((lane >> 4) & 3);
    const int rsub = lane >> 2;
    int apix[CFG::A_INS];
    unsigned tapmask[CFG::A_INS];
    const T* wrow[CFG::B_INS];
    {
        const int ph = p.KH / 2, pw = p.KW / 2;
#pragma unroll
        for (int i = 0; i < CFG::A_INS; ++i) {
            const long long m = m0 + wv * (BM / 4) + 16 * i + rsub;
            apix[i] = 0; tapmask[i] = 0u;
            if (m < M) {
                const unsigned mu = (unsigned)m;
                const unsigned t = mu / (unsigned)p.Wo;
                const int xo = (int)(mu - t * (unsigned)p.Wo);
                const unsigned n = t / (unsigned)p.Ho;
                const int y = (int)(t - n * (unsigned)p.Ho) * p.stride, x = xo * p.stride;
                apix[i] = ((int)n * p.H + y) * p.W + x;
                unsigned mk = 0u;
                for (int a = 0; a < p.KH; ++a)
                    for (int b = 0; b < p.KW; ++b) {
                        const int yy = y + a - ph, xx = x + b - pw;
                        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) mk |= 1u << (a * p.KW + b);
                    }
                tapmask[i] = mk;
            }
        }
        const T* wp = static_cast<const T*>(p.weight);
#pragma unroll
        for (int i = 0; i < CFG::B_INS; ++i) {
            const int co = n0 + wv * (BN / 4) + 16 * i + rsub;
            wrow[i] = co < p.Cout ? wp + (size_t)co * Ktot + q * VEC : nullptr;
        }
    }
    // one K cursor per plane: channel within the tap, tap coordinates of piece q of the NEXT tile to request
    KCursor<CFG::PBK> cur[KP];
#pragma unroll
    for (int pl = 0; pl < KP; ++pl) cur[pl].init(p, pl * CFG::PBK + q * VEC);
    const T* zp = static_cast<const T*>(p.zero);
    const T* const s0 = static_cast<const T*>(p.src[0]);
    const T* const s1 = static_cast<const T*>(p.src[1]);
    const T* const s2 = static_cast<const T*>(p.src[2]);
    const T* const s3 = static_cast<const T*>(p.src[3]);
    const int st0 = p.src_stride[0], st1 = p.src_stride[1], st2 = p.src_stride[2], st3 = p.src_stride[3];
    const int c0n = p.src_c[0], c1n = c0n + p.src_c[1], c2n = c1n + p.src_c[2];

    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    char* const ring = smem;
    const unsigned ring_addr = (unsigned)(unsigned long long)(__attribute__((address_space(3))) char*)smem;   // LDS byte address
    // request K tile kt into ring slot `slot` (kt may run past the end: everything then reads the zero block)
    auto request = [&](int kt, int slot) __attribute__((always_inline)) {
        char* stage = ring + (size_t)slot * CFG::STAGE_BYTES;
#pragma unroll
        for (int pl = 0; pl < KP; ++pl) {
            const bool kvalid = kt * CFG::BK + pl * CFG::PBK + q * VEC < Ktot;
            const int kcp = cur[pl].kc, kyp = cur[pl].ky, kxp = cur[pl].kx;
            const T* sp = s0;
            unsigned ss = (unsigned)st0, c = (unsigned)kcp;
            if (p.nsrc > 1) {
                const bool g0 = kcp >= c0n, g1 = kcp >= c1n, g2 = kcp >= c2n;
                const long long d1 = (const char*)s1 - (const char*)s0, d2 = (const char*)s2 - (const char*)s1, d3 = (const char*)s3 - (const char*)s2;
                sp = reinterpret_cast<const T*>((const char*)s0 + ((g0 ? d1 : 0) + (g1 ? d2 : 0) + (g2 ? d3 : 0)));
                ss = (unsigned)(st0 + (g0 ? st1 - st0 : 0) + (g1 ? st2 - st1 : 0) + (g2 ? st3 - st2 : 0));
                c = (unsigned)(kcp - ((g0 ? c0n : 0) + (g1 ? c1n - c0n : 0) + (g2 ? c2n - c1n : 0)));
            }
            const int tapoff = (kyp - p.KH / 2) * p.W + (kxp - p.KW / 2);
            const unsigned tapbit = kvalid ? 1u << (kyp * p.KW + kxp) : 0u;
            char* plane = stage + (size_t)pl * CFG::PLANE_BYTES;
#pragma unroll
            for (int i = 0; i < CFG::A_INS; ++i) {
                const unsigned e = __umul24((unsigned)(apix[i] + tapoff), ss) + c;
                const T* src = (tapmask[i] & tapbit) ? sp + e : zp;
                __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(plane + (wv * (BM / 4) + 16 * i) * 64), 16, 0, 0);
            }
            const int koff = kt * CFG::BK + pl * CFG::PBK;
#pragma unroll
            for (int i = 0; i < CFG::B_INS; ++i) {
                const T* src = (kvalid && wrow[i]) ? wrow[i] + koff : zp;
                __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(plane + (BM + wv * (BN / 4) + 16 * i) * 64), 16, 0, 0);
            }
            cur[pl].template advance<CFG::BK>(p);
        }
    };

    float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int hi = lane >> 5, l31 = lane & 31;
    const int sw = (l31 >> 2) & 3;                                // swizzle of this lane's fragment rows
    int issued = 0;
#pragma unroll 1
    for (; issued < NS - 1; ++issued) request(issued, issued);
    int slot_c = 0, slot_i = NS - 1;                             // ring slots: being multiplied / next to fill
#pragma unroll 1
    for (int kt = 0; kt < nkt; ++kt) {
        // tile kt has landed for this wave once at most NS-2 younger tiles are pending; the barrier publishes every wave's part
        // and guarantees that slot_i (multiplied in the previous iteration) is free
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * CFG::LPS) : "memory");
        __builtin_amdgcn_s_barrier();                            // plain barrier: no fence, the DMA ring is tracked by hand
        request(kt + NS - 1, slot_i);
        const unsigned stage = ring_addr + (unsigned)slot_c * CFG::STAGE_BYTES;
        // fragment reads run one k16 step ahead of the MFMAs (two register sets): the LDS latency of step s+1 hides under the
        // MFMAs of step s; lgkmcnt is counted (LDS returns in order), never 0 while a younger step is in flight
        constexpr int NSTEP = KP * CFG::PSTEPS;
        constexpr int RPS = (CFG::MT + CFG::NTL) * (sizeof(T) == 2 ? 1 : 2);      // ds_read_b128 per step
        RawFrag<T> xr[2][CFG::MT], wr[2][CFG::NTL];
        auto issue_step = [&](int s, int set) __attribute__((always_inline)) {
            const int pl = s / CFG::PSTEPS, st = s % CFG::PSTEPS;
            const unsigned plane = stage + (unsigned)pl * CFG::PLANE_BYTES;
            const unsigned a = plane + (unsigned)(wm * CFG::WM + l31) * 64;
            const unsigned b = plane + (unsigned)(BM + wn * CFG::WN + l31) * 64;
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i) issue_frag(xr[set][i], a + (unsigned)i * 32 * 64, st, hi, sw);
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j) issue_frag(wr[set][j], b + (unsigned)j * 32 * 64, st, hi, sw);
        };
        issue_step(0, 0);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const int set = s & 1;
            if (s + 1 < NSTEP) {
                issue_step(s + 1, set ^ 1);
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(RPS) : "memory");
            } else {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i) settle(xr[set][i]);
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j) settle(wr[set][j]);
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                for (int j = 0; j < CFG::NTL; ++j) mma32(acc[i][j], to_frag(wr[set][j]), to_frag(xr[set][i]));    // D[cout][pixel]
        }
        slot_c = slot_c + 1 == NS ? 0 : slot_c + 1;
        slot_i = slot_i + 1 == NS ? 0 : slot_i + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the tail requests (zero block) must land before LDS is reused
    __syncthreads();

    CoutRegs<CFG> bias;
    bias.load(p.bias, p.zero, p.Cout, n0, wn, lane);
    const LinearPix pix{m0, M};
    AuxRegs<CFG, T> aux;
    aux.prefetch(p, tid, n0, pix);
    stage_tile_act<CFG, T>(p, acc, Cs, bias, wm, wn, lane);
    __syncthreads();
    store_tile<CFG, T>(p, Cs, tid, n0, aux, pix);
}

// ---------------------------------------------------------------------------------------------------------------
// v3 "halo" kernel for spatial kernels (3x3, 3x1, 1x3, stride 1): the block owns a PATCH of 4 rows x 32 columns of output
// pixels.  v1/v2 re-gather the input for each of the KH*KW taps (im2col through the texture path: at 128x128 tiles the
// vector-memory pipe is as busy as the MFMA pipe).  Here the patch plus its halo ((4+KH-1) x (32+KW-1) pixels, 64 bytes ... 128
// bytes of channels per pixel) is loaded ONCE per 128-byte channel chunk; every tap then reads its MFMA operand from the same
// LDS tile at a constant row offset (a 32-pixel MFMA tile is one image row segment = 32 consecutive LDS rows: conflict-free).
// Input traffic through the memory pipe drops from KH*KW x 16 KB to 26-30 KB per chunk; the weight tile (BN x 128 bytes per tap)
// is double buffered exactly as in v1.  K order of the loop: (chunk, ky, kx) -- the packed weight stays (Cout, KH, KW, Cin).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BN_, int NWAVES_ = 4, int WGM_ = 2, int DW_ = 1>
struct ConvCfgH {
    static constexpr int DW = DW_;                       // weight tiles in flight (register ring): short grids (one block per CU) need > 1
    static constexpr int BM = 128, BN = BN_, PH = 4, PW = 32, WGM = WGM_, WGN = NWAVES_ / WGM_;
    static constexpr int NT = 64 * NWAVES_;              // threads per block (4 or 8 waves)
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int BK = 8 * VEC;                   // channels per chunk (128 bytes)
    static constexpr int RS = BK + VEC;                  // LDS row stride (elements): 144 bytes
    static constexpr int KSTEPS = BK / 16;
    static constexpr int WM = BM / WGM, WN = BN / WGN, MT = WM / 32, NTL = WN / 32;
    static constexpr int MAXHALO = (PH + 2) * (PW + 2);  // 204 halo pixels for 3x3
    static constexpr int RPI = NT / 8;                   // tile rows covered by one pass of the loader threads
    static constexpr int A_IT = (MAXHALO + RPI - 1) / RPI;        // 16-byte pieces per thread and chunk
    static constexpr int B_IT = (BN + RPI - 1) / RPI;
    static constexpr int CRS = BN + VEC;
    static constexpr size_t A_BYTES = (size_t)MAXHALO * RS * sizeof(T);
    static constexpr size_t B_BYTES = (size_t)2 * BN * RS * sizeof(T);
    static constexpr size_t STAGE_BYTES = (size_t)BM * CRS * sizeof(T);
    static constexpr size_t TILE_BYTES = A_BYTES + B_BYTES;
    static constexpr size_t LDS_BYTES = TILE_BYTES > STAGE_BYTES ? TILE_BYTES : STAGE_BYTES;
};

template <typename CFG, typename T>
__global__ __launch_bounds__(CFG::NT) void conv_halo_kernel(ConvArgs p, int tiles_x, int tiles_y) {
    constexpr int BN = CFG::BN, VEC = CFG::VEC, RS = CFG::RS, BK = CFG::BK, PH = CFG::PH, PW = CFG::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ah = reinterpret_cast<T*>(smem);                          // [halo pixels][RS]
    T* Bs = reinterpret_cast<T*>(smem + CFG::A_BYTES);           // [2][BN][RS]
    T* Cs = reinterpret_cast<T*>(smem);                          // [128][CRS]  (after the K loop)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv % CFG::WGM, wn = wv / CFG::WGM;
    const int hi = lane >> 5, l31 = lane & 31;
    // patch of this block
    int bx = blockIdx.x;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int n = bx / tiles_y;
    const int y0 = ty * PH, x0 = tx * PW;
    const int n0 = blockIdx.y * BN;
    const int HW_ = p.KW + PW - 1, HH_ = p.KH + PH - 1;          // halo extent
    const int nhalo = HW_ * HH_;
    const int ph = p.KH / 2, pw = p.KW / 2;
    const int ntap = p.KH * p.KW;
    const int nchunk = (p.Cin + BK - 1) / BK;
    const int nkt = nchunk * ntap;
    const int Ktot = ntap * p.Cin;

    // ---- loader state
    const int pc = tid & 7;
    int apix[CFG::A_IT];                                          // input pixel index of this thread's halo pixels, -1: outside / unused
#pragma unroll
    for (int it = 0; it < CFG::A_IT; ++it) {
        const int hp = (tid >> 3) + CFG::RPI * it;
        apix[it] = -1;
        if (hp < nhalo) {
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int yy = y0 - ph + hy, xx = x0 - pw + hx;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) apix[it] = (n * p.H + yy) * p.W + xx;
        }
    }
    const T* wrow[CFG::B_IT];
    {
        const T* wp = static_cast<const T*>(p.weight);
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) {
            const int br = (tid >> 3) + CFG::RPI * it;
            const int co = n0 + br;
            wrow[it] = (br < BN && co < p.Cout) ? wp + (size_t)co * Ktot + pc * VEC : nullptr;
        }
    }
    const T* zp = static_cast<const T*>(p.zero);
    const T* const s0 = static_cast<const T*>(p.src[0]);
    const T* const s1 = static_cast<const T*>(p.src[1]);
    const T* const s2 = static_cast<const T*>(p.src[2]);
    const T* const s3 = static_cast<const T*>(p.src[3]);
    const int st0 = p.src_stride[0], st1 = p.src_stride[1], st2 = p.src_stride[2], st3 = p.src_stride[3];
    const int c0n = p.src_c[0], c1n = c0n + p.src_c[1], c2n = c1n + p.src_c[2];

    raw16_t ra[CFG::A_IT], rb[CFG::DW][CFG::B_IT];
    auto fetch_a = [&](int chunk) __attribute__((always_inline)) {            // halo tile of one channel chunk -> registers
        const int kc = chunk * BK + pc * VEC;
        const bool cvalid = kc < p.Cin;
        const T* sp = s0;
        unsigned ss = (unsigned)st0, c = (unsigned)kc;
        if (p.nsrc > 1) {
            const bool g0 = kc >= c0n, g1 = kc >= c1n, g2 = kc >= c2n;
            const long long d1 = (const char*)s1 - (const char*)s0, d2 = (const char*)s2 - (const char*)s1, d3 = (const char*)s3 - (const char*)s2;
            sp = reinterpret_cast<const T*>((const char*)s0 + ((g0 ? d1 : 0) + (g1 ? d2 : 0) + (g2 ? d3 : 0)));
            ss = (unsigned)(st0 + (g0 ? st1 - st0 : 0) + (g1 ? st2 - st1 : 0) + (g2 ? st3 - st2 : 0));
            c = (unsigned)(kc - ((g0 ? c0n : 0) + (g1 ? c1n - c0n : 0) + (g2 ? c2n - c1n : 0)));
        }
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const unsigned e = __umul24((unsigned)apix[it], ss) + c;
            const T* src = (cvalid && apix[it] >= 0) ? sp + e : zp;
            ra[it] = global_load16(src);
        }
    };
    auto stash_a = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const int hp = (tid >> 3) + CFG::RPI * it;
            if (hp < CFG::MAXHALO) *reinterpret_cast<raw16_t*>(Ah + (size_t)hp * RS + pc * VEC) = ra[it];
        }
    };
    auto fetch_b = [&](int kt, int slot) __attribute__((always_inline)) {     // weight tile of K tile kt = (chunk, tap)
        const int chunk = kt / ntap, tap = kt - chunk * ntap;
        const bool cvalid = chunk * BK + pc * VEC < p.Cin;
        const int koff = tap * p.Cin + chunk * BK;
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) {
            const T* src = (cvalid && wrow[it]) ? wrow[it] + koff : zp;
            rb[slot][it] = global_load16(src);
        }
    };
    auto stash_b = [&](int buf, int slot) __attribute__((always_inline)) {
        T* b = Bs + (size_t)buf * BN * RS;
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) {
            const int br = (tid >> 3) + CFG::RPI * it;
            if (br < BN) *reinterpret_cast<raw16_t*>(b + (size_t)br * RS + pc * VEC) = rb[slot][it];
        }
    };

    CoutRegs<CFG> bias;                                           // requested now, used after the K loop
    bias.load(p.bias, p.zero, p.Cout, n0, wn, lane);
    float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
        for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    constexpr int DW = CFG::DW;
    if constexpr (DW == 1) {
        // one weight tile ahead, compiler-tracked loads (the small 4-wave tiles: several blocks per CU cover each other's latency)
        fetch_a(0);
        fetch_b(0, 0);
        stash_a();
        stash_b(0, 0);
        __syncthreads();
        int tap = 0, chunk = 0;
#pragma unroll 1
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = kt & 1;
            const bool more = kt + 1 < nkt;
            const bool last_tap = tap == ntap - 1;
            const bool next_chunk = last_tap && chunk + 1 < nchunk;
            if (more) fetch_b(kt + 1, 0);                             // in flight under the MFMAs
            if (tap == 0 && chunk + 1 < nchunk) fetch_a(chunk + 1);   // next halo tile: requested now, parked in registers until the last tap
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const T* a = Ah + (size_t)((wm * CFG::MT + ky) * HW_ + l31 + kx) * RS + hi * 8;
            const T* b = Bs + (size_t)buf * BN * RS + (size_t)(wn * CFG::WN + l31) * RS + hi * 8;
#pragma unroll
            for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
                Frag<T> xf[CFG::MT], wf[CFG::NTL];
#pragma unroll
                for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], a + (size_t)i * HW_ * RS + kk * 16);
#pragma unroll
                for (int j = 0; j < CFG::NTL; ++j) load_frag(wf[j], b + (size_t)j * 32 * RS + kk * 16);
#pragma unroll
                for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                    for (int j = 0; j < CFG::NTL; ++j) mma32(acc[i][j], wf[j], xf[i]);    // D[cout][pixel]
            }
            if (more) stash_b(buf ^ 1, 0);
            if (next_chunk) {
                __syncthreads();                                      // every wave is done with this chunk's halo tile
                stash_a();
            }
            __syncthreads();
            if (last_tap) { tap = 0; ++chunk; } else ++tap;
        }
    } else {
        // DW weight tiles in flight in a ring of register slots (slot = tile % DW, static after unrolling), all loads of the loop
        // UNTRACKED (global_load16_async, common.h) with counted waits: every iteration issues exactly B_IT weight loads (tiles past
        // the end read the zero page) and stashes exactly one tile, so "tile kt+1 has landed" is `vmcnt <= (DW-1)*B_IT` -- plus A_IT
        // while the halo loads of the next chunk (issued at tap 0, consumed at the last tap) are younger than that tile.
        constexpr bool HA = (S2M2_ASYNC_LOADS & 1) != 0;
        constexpr int NB = (DW - 1) * CFG::B_IT;
        int f_tap = 0, f_chunk = 0;                                   // K position of the next weight tile to request
        auto fetch_b_async = [&](int slot) __attribute__((always_inline)) {
            const bool cvalid = f_chunk < nchunk && f_chunk * BK + pc * VEC < p.Cin;
            const int koff = f_tap * p.Cin + f_chunk * BK;
#pragma unroll
            for (int it = 0; it < CFG::B_IT; ++it) {
                const T* src = (cvalid && wrow[it]) ? wrow[it] + koff : zp;
                global_load16_async<HA>(rb[slot][it], src);
            }
            if (++f_tap == ntap) { f_tap = 0; ++f_chunk; }
        };
        auto stash_b_async = [&](int buf, int slot) __attribute__((always_inline)) {
            T* b = Bs + (size_t)buf * BN * RS;
#pragma unroll
            for (int it = 0; it < CFG::B_IT; ++it) {
                settle(rb[slot][it]);
                const int br = (tid >> 3) + CFG::RPI * it;
                if (br < BN) *reinterpret_cast<raw16_t*>(b + (size_t)br * RS + pc * VEC) = rb[slot][it];
            }
        };
        auto fetch_a_async = [&](int chunk) __attribute__((always_inline)) {
            const int kc = chunk * BK + pc * VEC;
            const bool cvalid = kc < p.Cin;
            const T* sp = s0;
            unsigned ss = (unsigned)st0, c = (unsigned)kc;
            if (p.nsrc > 1) {
                const bool g0 = kc >= c0n, g1 = kc >= c1n, g2 = kc >= c2n;
                const long long d1 = (const char*)s1 - (const char*)s0, d2 = (const char*)s2 - (const char*)s1, d3 = (const char*)s3 - (const char*)s2;
                sp = reinterpret_cast<const T*>((const char*)s0 + ((g0 ? d1 : 0) + (g1 ? d2 : 0) + (g2 ? d3 : 0)));
                ss = (unsigned)(st0 + (g0 ? st1 - st0 : 0) + (g1 ? st2 - st1 : 0) + (g2 ? st3 - st2 : 0));
                c = (unsigned)(kc - ((g0 ? c0n : 0) + (g1 ? c1n - c0n : 0) + (g2 ? c2n - c1n : 0)));
            }
#pragma unroll
            for (int it = 0; it < CFG::A_IT; ++it) {
                const unsigned e = __umul24((unsigned)apix[it], ss) + c;
                const T* src = (cvalid && apix[it] >= 0) ? sp + e : zp;
                global_load16_async<HA>(ra[it], src);
            }
        };
        auto stash_a_async = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < CFG::A_IT; ++it) {
                settle(ra[it]);
                const int hp = (tid >> 3) + CFG::RPI * it;
                if (hp < CFG::MAXHALO) *reinterpret_cast<raw16_t*>(Ah + (size_t)hp * RS + pc * VEC) = ra[it];
            }
        };

        // (the tracked bias loads issued above are older than everything below: they only make the counted waits conservative)
        fetch_a_async(0);
#pragma unroll
        for (int f = 0; f < DW; ++f) fetch_b_async(f);
        wait_vmcnt<DW * CFG::B_IT, HA>();
        stash_a_async();
        wait_vmcnt<NB, HA>();
        stash_b_async(0, 0);
        __syncthreads();
        int tap = 0, chunk = 0, ky = 0, kx = 0;
#pragma unroll 1
        for (int kt0 = 0; kt0 < nkt; kt0 += DW) {
#pragma unroll
            for (int f = 0; f < DW; ++f) {
                const int kt = kt0 + f;
                if (kt >= nkt) break;
                const int buf = kt & 1;
                const bool last_tap = tap == ntap - 1;
                const bool has_next = chunk + 1 < nchunk;
                if (tap == 0 && has_next) fetch_a_async(chunk + 1);   // next halo tile: parked in registers until the last tap
                fetch_b_async(f);                                     // tile kt + DW into the slot stashed one tile ago
                const T* a = Ah + (size_t)((wm * CFG::MT + ky) * HW_ + l31 + kx) * RS + hi * 8;
                const T* b = Bs + (size_t)buf * BN * RS + (size_t)(wn * CFG::WN + l31) * RS + hi * 8;
#pragma unroll
                for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
                    Frag<T> xf[CFG::MT], wf[CFG::NTL];
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], a + (size_t)i * HW_ * RS + kk * 16);
#pragma unroll
                    for (int j = 0; j < CFG::NTL; ++j) load_frag(wf[j], b + (size_t)j * 32 * RS + kk * 16);
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                        for (int j = 0; j < CFG::NTL; ++j) mma32(acc[i][j], wf[j], xf[i]);    // D[cout][pixel]
                }
                if (has_next && tap <= DW - 2) wait_vmcnt<NB + CFG::A_IT, HA>();   // the halo loads of the next chunk are younger than tile kt+1
                else wait_vmcnt<NB, HA>();
                stash_b_async(buf ^ 1, (f + 1) % DW);                 // (past the end: a zero tile into the idle buffer)
                if (last_tap && has_next) {
                    __syncthreads();                                  // every wave is done with this chunk's halo tile
                    stash_a_async();                                  // older than every load still allowed in flight (ntap >= DW - 1)
                }
                __syncthreads();
                if (last_tap) { tap = 0; ky = 0; kx = 0; ++chunk; }
                else { ++tap; if (++kx == p.KW) { kx = 0; ++ky; } }
            }
        }
        wait_vmcnt<0, HA>();                                              // drain the zero-page tail requests before their registers are reused
#pragma unroll
        for (int f = 0; f < DW; ++f)
#pragma unroll
            for (int it = 0; it < CFG::B_IT; ++it) settle(rb[f][it]);
    }

    // ---- epilogues: staging rows r = patch row * 32 + column (the same wm*64 + i*32 + lane map as the linear kernels)
    const PatchPix pix{n, y0, x0, p.H, p.W};
    AuxRegs<CFG, T> aux;
    aux.prefetch(p, tid, n0, pix);
    stage_tile_act<CFG, T>(p, acc, Cs, bias, wm, wn, lane);
    __syncthreads();
    store_tile<CFG, T>(p, Cs, tid, n0, aux, pix);
}

// ---------------------------------------------------------------------------------------------------------------
// v5 "fragment stream" kernel for the spatial kernels (3x3, 3x1, 1x3, stride 1) of wide layers (fp16, Cin a multiple of 64,
// Cout a multiple of BN).  v3 stages BOTH operands of every K tile through LDS behind one block barrier per tile: its 8 waves
// read 1.5 KB of LDS per MFMA (192 B/clk/CU at the MFMA rate, the LDS delivers 256 at best) and meet at 18+ barriers per block.
// Here
//   * the weights never touch LDS: they are packed once (s2m2_amd/pack.py: pack_conv_frag, K order 2) as a stream of 1 KB
//     MFMA A-fragments per 32-cout tile in exactly the order the K loop consumes them -- (channel chunk, tap, k16 step) -- so a
//     wave's weight traffic is one perfectly coalesced, L2-resident stream, prefetched 8 fragments (= one tap) ahead in registers
//     with untracked loads and counted waits (common.h);
//   * the block loads the patch + halo for ALL channels of a chunk (128) at once: after that barrier the K loop of the chunk
//     (taps x 8 k16 steps) has no block-level synchronisation at all;
//   * a wave owns the whole 128-pixel patch x 32 couts (MT = 4, NTL = 1): one weight fragment feeds 4 MFMAs, LDS traffic is
//     1 KB per MFMA ... 128 B/clk/CU at the full MFMA rate from 4 waves, weights 32 B/clk/CU from L2.
// K order of the accumulation: (chunk, tap, channel) -- fp32 accumulate, so only the summation order differs from v3.
// ---------------------------------------------------------------------------------------------------------------
// timeline instrumentation of conv_frag_kernel for tools/frag_trace.py (experiment builds only: -DS2M2_FRAG_TRACE=1)
#ifndef S2M2_FRAG_TRACE
#define S2M2_FRAG_TRACE 0
#endif
#if S2M2_FRAG_TRACE
// slots 0..6: shader-clock stamps (s_memtime) of the kernel's phases; 7 / 8: the 100 MHz real-time counter (one clock for the whole chip) at entry / exit;
// 9: HW_ID (CU / SE / wave slot) | XCC_ID << 32
__device__ unsigned long long g_frag_trace[4096 * 4 * 10];
#define FRAG_T(slot)                                                                                                          \
    do {                                                                                                                      \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 4096 && blockIdx.y == 0) {                                                \
            unsigned long long* tr_ = g_frag_trace + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 10;                     \
            tr_[(slot)] = __builtin_amdgcn_s_memtime();                                                                       \
            if ((slot) == 0) {                                                                                                \
                tr_[7] = __builtin_amdgcn_s_memrealtime();                                                                    \
                tr_[9] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |                    \
                         ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);             \
            }                                                                                                                 \
            if ((slot) == 6) tr_[8] = __builtin_amdgcn_s_memrealtime();                                                       \
        }                                                                                                                     \
    } while (0)
#else
#define FRAG_T(slot)
#endif

template <typename T, int BN_, int CH_, int PH_ = 4, int PW_ = 32>
struct ConvCfgF {
    static_assert(sizeof(T) == 2, "the fragment-stream kernel is fp16 only");
    // PH: patch rows (4: 128 pixels per block; 2: 64 pixels -- twice the blocks on the coarse pyramid levels, whose grids do not fill the chip)
    // PW: patch columns.  32: an MFMA pixel tile is a patch row.  40 (PH = 4: 160 pixels, 5 MFMA tiles that run across patch rows): picked
    // by the launcher where it removes a partial round of blocks -- 256x304 is 640 patches of 4x32 on 512 resident block slots (2 per CU),
    // i.e. two rounds the second of which runs at 25 % occupancy, but 512 patches of 4x40: one round
    static constexpr int BM = PW_ * PH_, BN = BN_, PH = PH_, PW = PW_, WGM = 1, WGN = BN_ / 32;
    static constexpr int NWAVES = WGN, NT = 64 * NWAVES;
    static constexpr int VEC = 8, CH = CH_, KS = CH_ / 16;       // channels per chunk, k16 steps per tap
    static constexpr int RS = CH + VEC;                  // LDS row stride (elements): 16 bytes of padding per halo pixel
    static constexpr int WM = BM, WN = 32, MT = BM / 32, NTL = 1;
    static_assert(BM % 32 == 0, "a patch is a whole number of 32-pixel MFMA tiles");
    static constexpr int MAXHALO = (PH + 2) * (PW + 2);
    static constexpr int PPX = CH / VEC;                 // 16-byte pieces per halo pixel
    static constexpr int RPI = NT / PPX;                 // halo pixels covered by one pass of the loader threads
    static constexpr int A_IT = (MAXHALO + RPI - 1) / RPI;
    static constexpr int CRS = BN + VEC;
    static constexpr int AROWS = A_IT * RPI;             // LDS rows: every loader thread stashes all of its pieces, no exec-masked tail
    static constexpr size_t A_BYTES = (size_t)AROWS * RS * sizeof(T);
    static constexpr size_t STAGE_BYTES = (size_t)BM * CRS * sizeof(T);
    static constexpr size_t LDS_BYTES = A_BYTES > STAGE_BYTES ? A_BYTES : STAGE_BYTES;
    static_assert(NT % PPX == 0 && KS >= 2 && KS % 2 == 0, "loader geometry");
};

// AUX: epilogue operands parked in registers through the K loop: 0 none (epi == NONE), 1 one (add / mul), 2 two (GRU blend, gate mix)
template <typename CFG, typename T, int AUX>
__global__ __launch_bounds__(CFG::NT) void conv_frag_kernel(ConvArgs p, int tiles_x, int tiles_y) {
    constexpr int BN = CFG::BN, VEC = CFG::VEC, RS = CFG::RS, CH = CFG::CH, KS = CFG::KS, PH = CFG::PH, PW = CFG::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ah = reinterpret_cast<T*>(smem);                          // [halo pixels][RS]
    T* Cs = reinterpret_cast<T*>(smem);                          // [128][CRS]  (after the K loop)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // = cout tile of this wave inside the block
    const int hi = lane >> 5, l31 = lane & 31;
    // patch <-> block id exactly as in v3 (hardware: block b runs on XCD b % 8): the layers before and after this one then find / leave
    // every patch in the L2 of the XCD that works on it next.  (An XCD-contiguous remap of the ids made THIS kernel's halo reads
    // cheaper and the next layer's reads miss: measured +21 us per pass on the second conv of the ConvBlock2D pairs.)
    int bx = blockIdx.x;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int n = bx / tiles_y;
    const int y0 = ty * PH, x0 = tx * PW;
    const int n0 = blockIdx.y * BN;
    const int HW_ = p.KW + PW - 1, HH_ = p.KH + PH - 1;          // halo extent
    const int nhalo = HW_ * HH_;
    const int ph = p.KH / 2, pw = p.KW / 2;
    const int ntap = p.KH * p.KW;
    const int nchunk = (p.Cin + CH - 1) / CH;
    const int nfrag = nchunk * ntap * KS;                         // fragments of this wave's stream
    FRAG_T(0);
    // ---- halo loader state (as in v3, CH channels per pixel)
    const int pc = tid % CFG::PPX, prow = tid / CFG::PPX;
    int apix[CFG::A_IT];
#pragma unroll
    for (int it = 0; it < CFG::A_IT; ++it) {
        const int hp = prow + CFG::RPI * it;
        apix[it] = -1;
        if (hp < nhalo) {
            const int hy = hp / HW_, hx = hp - hy * HW_;
            const int yy = y0 - ph + hy, xx = x0 - pw + hx;
            if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) apix[it] = (n * p.H + yy) * p.W + xx;
        }
    }
    const T* zp = static_cast<const T*>(p.zero);
    const T* const s0 = static_cast<const T*>(p.src[0]);
    const T* const s1 = static_cast<const T*>(p.src[1]);
    const T* const s2 = static_cast<const T*>(p.src[2]);
    const T* const s3 = static_cast<const T*>(p.src[3]);
    const int st0 = p.src_stride[0], st1 = p.src_stride[1], st2 = p.src_stride[2], st3 = p.src_stride[3];
    const int c0n = p.src_c[0], c1n = c0n + p.src_c[1], c2n = c1n + p.src_c[2];
    auto load_halo = [&](int chunk) __attribute__((always_inline)) {          // patch + halo of one channel chunk -> LDS
        const int kc = chunk * CH + pc * VEC;
        const bool cvalid = kc < p.Cin;
        const T* sp = s0;
        unsigned ss = (unsigned)st0, c = (unsigned)kc;
        if (p.nsrc > 1) {
            const bool g0 = kc >= c0n, g1 = kc >= c1n, g2 = kc >= c2n;
            const long long d1 = (const char*)s1 - (const char*)s0, d2 = (const char*)s2 - (const char*)s1, d3 = (const char*)s3 - (const char*)s2;
            sp = reinterpret_cast<const T*>((const char*)s0 + ((g0 ? d1 : 0) + (g1 ? d2 : 0) + (g2 ? d3 : 0)));
            ss = (unsigned)(st0 + (g0 ? st1 - st0 : 0) + (g1 ? st2 - st1 : 0) + (g2 ? st3 - st2 : 0));
            c = (unsigned)(kc - ((g0 ? c0n : 0) + (g1 ? c1n - c0n : 0) + (g2 ? c2n - c1n : 0)));
        }
        raw16_t ra[CFG::A_IT];
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const unsigned e = __umul24((unsigned)apix[it], ss) + c;
            const T* src = (cvalid && apix[it] >= 0) ? sp + e : zp;
            ra[it] = global_load16(src);
        }
        // every request is consumed, unconditionally (rows past the halo are padding): a tracked load whose use sits under an exec
        // mask stays "pending" in the compiler's bookkeeping and turns into an s_waitcnt vmcnt(0) at the top of the tap loop, which
        // would drain the untracked weight ring on every tap
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it)
            *reinterpret_cast<raw16_t*>(Ah + (size_t)(prow + CFG::RPI * it) * RS + pc * VEC) = ra[it];
    };

    // ---- weight fragment stream of this wave: fragment f at wf + f * 64 (16-byte units), one 16-byte piece per lane
    const raw16_t* wf = static_cast<const raw16_t*>(p.weight) + ((size_t)(blockIdx.y * CFG::WGN + wv) * nfrag) * 64 + lane;
    raw16_t ring[KS];

    CoutRegs<CFG> bias;                                           // requested now, used after the K loop
    bias.load(p.bias, p.zero, p.Cout, n0, wv, lane);
    // epilogue operand (residual / gate, one-operand epilogues): requested FIRST, as ordinary tracked loads, and parked in 32 registers
    // through the K loop.  Being older than every ring request they never enter the ring's counted waits (requests complete in
    // order; a tracked load the compiler moved below the ring's first requests would only make those waits conservative), and
    // their latency runs beside the halo tile's.  (Requested after the K loop, 8 pieces per thread sat in front of the stores:
    // +3.5 us per block, profiles/r02/frag_timeline.txt.)
    const PatchPixT<PW> pix{n, y0, x0, p.H, p.W};
    // 160-pixel blocks (PW = 40): 10 operand pieces per thread do not fit next to the K loop's registers at two blocks per CU -- they are
    // requested AFTER the K loop (the ring and the pixel fragments are dead by then) and arrive under the bias / activation / staging pass
    constexpr bool AUX_LATE = AUX != 0 && PW != 32;
    using AX = AuxRegs<CFG, T, AUX != 0 ? (AUX_LATE ? 10 : 8) : 4>;
    AX aux;
    // stacked layers with different epilogues (epi_cout0): cout blocks below it store bias + activation only
    const int epi_eff = (n0 < p.epi_cout0) ? (int)S2M2_EPI_NONE : p.epi;
    if constexpr (AUX != 0 && !AUX_LATE) aux.template prefetch<AUX == 1>(p, tid, n0, pix, epi_eff);
    float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;

    // the first tap's fragments are requested before the halo tile: both latencies run together (the tracked halo loads are
    // younger, so the compiler's waits at the stash cover the ring as well)
#pragma unroll
    for (int s = 0; s < KS; ++s) global_load16_async(ring[s], wf + (size_t)(s < nfrag ? s : nfrag - 1) * 64);
    load_halo(0);
    __syncthreads();
    FRAG_T(1);
    int g = 0;                                                    // global k16 step = index of the fragment consumed next
    // halo-tile offset (elements) of this lane's pixel in MFMA tile i: pixel q = 32 i + lane % 32 of the patch in raster order
    int poff[CFG::MT];
#pragma unroll
    for (int i = 0; i < CFG::MT; ++i) {
        if constexpr (PW == 32) poff[i] = (i * HW_ + l31) * RS;
        else { const int q = 32 * i + l31, qy = q / PW; poff[i] = (qy * HW_ + (q - qy * PW)) * RS; }
    }
    auto tap_steps = [&](int ky, int kx) __attribute__((always_inline)) {          // the KS k16 steps of one tap
        const T* a = Ah + (size_t)(ky * HW_ + kx) * RS + hi * 8;
        Frag<T> xf[2][CFG::MT];                                   // pixel fragments, double buffered across k16 steps
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i) load_frag(xf[0][i], a + poff[i]);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            if (kk + 1 < KS) {
#pragma unroll
                for (int i = 0; i < CFG::MT; ++i) load_frag(xf[(kk + 1) & 1][i], a + poff[i] + (kk + 1) * 16);
            }
            // fragment g (slot kk) was requested KS - 1 steps ago; the KS - 2 requests made since then may still be in flight
            wait_vmcnt<KS - 2>();
            settle(ring[kk]);
            Frag<T> wfr;
            wfr.v = __builtin_bit_cast(half8_t, ring[kk]);
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i) mma32(acc[i][0], wfr, xf[kk & 1][i]);   // D[cout][pixel]
            // refill the slot consumed one step ago (its MFMAs have long read their operands) with fragment g + KS - 1.
            // UNCONDITIONAL: a branch around an untracked load makes the compiler merge the two register states with copies
            // that read the slot while its load is in flight.  (Step 0 re-requests fragment KS - 1 into its own slot.)
            {
                const int f = g + KS - 1;
                global_load16_async(ring[(kk + KS - 1) % KS], wf + (size_t)(f < nfrag ? f : nfrag - 1) * 64);
            }
            ++g;
        }
    };
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        if (chunk > 0) {
            wait_vmcnt<0>();                                      // ring loads land before tracked loads are mixed in (their data stays valid)
            __syncthreads();                                      // every wave is done with the previous chunk's halo tile
            load_halo(chunk);
            __syncthreads();
        }
        int ky = 0, kx = 0;
#pragma unroll 1
        for (int tap = 0; tap < ntap; ++tap) {
            tap_steps(ky, kx);
            if (++kx == p.KW) { kx = 0; ++ky; }
        }
    }
    wait_vmcnt<0>();                                              // drain the tail requests before their registers are reused
#pragma unroll
    for (int s = 0; s < KS; ++s) settle(ring[s]);

    // ---- epilogues: staging rows r = patch pixel in raster order
    if constexpr (AUX_LATE) aux.template prefetch<AUX == 1>(p, tid, n0, pix, epi_eff);
    FRAG_T(2);
    __syncthreads();                                              // the staging tile aliases the halo tile
    FRAG_T(3);
    stage_tile_act<CFG, T>(p, acc, Cs, bias, 0, wv, lane);
    FRAG_T(4);
    __syncthreads();
    FRAG_T(5);
    store_tile<CFG, T>(p, Cs, tid, n0, aux, pix, epi_eff);        // (!AUX: launched with epi == NONE only, no operand is read)
    FRAG_T(6);
}

// ---------------------------------------------------------------------------------------------------------------
// v4 "pointwise" kernel for 1x1 convolutions / linear layers (K = Cin <= 512): these are streaming, HBM-bound GEMMs (M ~ 10^5
// pixels, K and N a few hundred) where the tiled kernels spend most of their time in per-block prologues: every 64x64 tile
// re-loads its weight slice and drains its two-tile pipeline.  Here a PERSISTENT block keeps the weight slice of its BN output
// channels in LDS for its whole life and walks over pixel tiles of 64: the activation stream (64 px x 128 bytes per step, double
// buffered through registers) never stops at tile boundaries, the epilogue of tile i (staging tile separate from the ring) runs
// under the loads of tile i+1, and the memory pipe carries activations only.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int BN_>
struct ConvCfgP {
    static constexpr int BM = 64, BN = BN_, WGM = 2, WGN = 2, NT = 256;
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int BK = 8 * VEC;                   // K elements per step (128 bytes)
    static constexpr int RS = BK + VEC;                  // activation row stride in LDS (elements)
    static constexpr int KSTEPS = BK / 16;
    static constexpr int WM = 32, WN = BN / 2, MT = 1, NTL = WN / 32;
    static constexpr int A_IT = 2;                       // 64 rows x 8 pieces / 256 threads
    static constexpr int CRS = BN + VEC;
    static constexpr size_t A_BYTES = (size_t)2 * BM * RS * sizeof(T);
    static constexpr size_t C_BYTES = (size_t)BM * CRS * sizeof(T);
    static size_t w_bytes(int K) { return (size_t)BN * (K + VEC) * sizeof(T); }
    static size_t lds_bytes(int K) { return A_BYTES + C_BYTES + w_bytes(K); }
};

template <typename CFG, typename T>
__global__ __launch_bounds__(256) void conv_pw_kernel(ConvArgs p, int ntiles) {
    constexpr int BM = CFG::BM, BN = CFG::BN, VEC = CFG::VEC, RS = CFG::RS, BK = CFG::BK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);                                  // [2][BM][RS]
    T* Cs = reinterpret_cast<T*>(smem + CFG::A_BYTES);                   // [BM][CRS]
    T* Ws = reinterpret_cast<T*>(smem + CFG::A_BYTES + CFG::C_BYTES);    // [BN][K + VEC]
    const int K = p.Cin;
    const int WRS = K + VEC;
    const int nchunk = (K + BK - 1) / BK;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wv & 1, wn = wv >> 1;
    const int hi = lane >> 5, l31 = lane & 31;
    const long long M = (long long)p.N * p.H * p.W;
    const int n0 = blockIdx.y * BN;
    CoutRegs<CFG> bias;                                           // the block keeps its cout slice for its whole life
    bias.load(p.bias, p.zero, p.Cout, n0, wn, lane);
    const T* zp = static_cast<const T*>(p.zero);

    // ---- weight slice -> LDS once (rows past Cout are zero)
    {
        const T* wp = static_cast<const T*>(p.weight);
        const int ppr = K / VEC;                                          // pieces per weight row
        for (int q = tid; q < BN * ppr; q += 256) {
            const int r = q / ppr, pc = q - r * ppr;
            const int co = n0 + r;
            const T* src = co < p.Cout ? wp + (size_t)co * K + pc * VEC : zp;
            *reinterpret_cast<raw16_t*>(Ws + (size_t)r * WRS + pc * VEC) = global_load16(src);
        }
    }
    // ---- activation loader: piece column pc of rows lrow, lrow + 32 of the current pixel tile
    const int pc = tid & 7, lrow = tid >> 3;
    const T* const s0 = static_cast<const T*>(p.src[0]);
    const T* const s1 = static_cast<const T*>(p.src[1]);
    const T* const s2 = static_cast<const T*>(p.src[2]);
    const T* const s3 = static_cast<const T*>(p.src[3]);
    const int st0 = p.src_stride[0], st1 = p.src_stride[1], st2 = p.src_stride[2], st3 = p.src_stride[3];
    const int c0n = p.src_c[0], c1n = c0n + p.src_c[1], c2n = c1n + p.src_c[2];
    raw16_t ra[CFG::A_IT];
    auto fetch = [&](long long m0, int chunk) __attribute__((always_inline)) {
        const int kc = chunk * BK + pc * VEC;
        const bool cvalid = kc < K;
        const T* sp = s0;
        unsigned ss = (unsigned)st0, c = (unsigned)kc;
        if (p.nsrc > 1) {
            const bool g0 = kc >= c0n, g1 = kc >= c1n, g2 = kc >= c2n;
            const long long d1 = (const char*)s1 - (const char*)s0, d2 = (const char*)s2 - (const char*)s1, d3 = (const char*)s3 - (const char*)s2;
            sp = reinterpret_cast<const T*>((const char*)s0 + ((g0 ? d1 : 0) + (g1 ? d2 : 0) + (g2 ? d3 : 0)));
            ss = (unsigned)(st0 + (g0 ? st1 - st0 : 0) + (g1 ? st2 - st1 : 0) + (g2 ? st3 - st2 : 0));
            c = (unsigned)(kc - ((g0 ? c0n : 0) + (g1 ? c1n - c0n : 0) + (g2 ? c2n - c1n : 0)));
        }
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const long long m = m0 + lrow + 32 * it;
            const T* src = (cvalid && m < M) ? sp + (unsigned)m * ss + c : zp;
            ra[it] = global_load16(src);
        }
    };
    auto stash = [&](int buf) __attribute__((always_inline)) {
        T* a = As + (size_t)buf * BM * RS;
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) *reinterpret_cast<raw16_t*>(a + (size_t)(lrow + 32 * it) * RS + pc * VEC) = ra[it];
    };

    float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
    for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

    // flat walk over (tile, chunk): step s -> tile blockIdx.x + (s / nchunk) * gridDim.x
    const int mytiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nstep = mytiles * nchunk;
    if (nstep <= 0) return;
    long long mt_cur = blockIdx.x;                                        // tile of the step being multiplied
    int chunk_cur = 0;
    long long mt_nxt = blockIdx.x;                                        // tile / chunk of the step being fetched
    int chunk_nxt = 0;
    fetch(mt_nxt * BM, chunk_nxt);
    stash(0);
    __syncthreads();                                                      // also publishes the weight slice
#pragma unroll 1
    for (int s = 0; s < nstep; ++s) {
        const int buf = s & 1;
        if (++chunk_nxt == nchunk) { chunk_nxt = 0; mt_nxt += gridDim.x; }
        const bool more = s + 1 < nstep;
        if (more) fetch(mt_nxt * BM, chunk_nxt);                          // in flight under the MFMAs and the epilogue
        const T* a = As + (size_t)buf * BM * RS + (size_t)(wm * 32 + l31) * RS + hi * 8;
        const T* b = Ws + (size_t)(wn * CFG::WN + l31) * WRS + chunk_cur * BK + hi * 8;
#pragma unroll
        for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
            Frag<T> xf, wf[CFG::NTL];
            load_frag(xf, a + kk * 16);
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j) load_frag(wf[j], b + (size_t)j * 32 * WRS + kk * 16);
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j) mma32(acc[0][j], wf[j], xf);      // D[cout][pixel]
        }
        if (chunk_cur == nchunk - 1) {
            // ---- tile finished: epilogue (Cs is separate from the ring; the k-step barriers order its reuse)
            stage_tile_act<CFG, T>(p, acc, Cs, bias, wm, wn, lane);
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
            if (more) stash(buf ^ 1);
            __syncthreads();
            {
                const LinearPix pix{mt_cur * BM, M};
                AuxRegs<CFG, T> aux;
                aux.prefetch(p, tid, n0, pix);
                store_tile<CFG, T>(p, Cs, tid, n0, aux, pix);
            }
            mt_cur += gridDim.x;
            chunk_cur = 0;
        } else {
            if (more) stash(buf ^ 1);
            __syncthreads();
            ++chunk_cur;
        }
    }
}

template <typename T, int BM, int BN, int WGM, int PPR = 8, int NPF = 1, int NWAVES = 4, int MODE = 0>
static int launch_conv(const ConvArgs& a, hipStream_t st) {
    using CFG = ConvCfg<T, BM, BN, WGM, PPR, NPF, NWAVES>;
    auto kern = conv_igemm_kernel<CFG, T, MODE>;
    static size_t lds_granted[kMaxDevices] = {};                     // per instantiation
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted, "conv2d")) return 1;
    const long long M = (long long)a.N * a.Ho * a.Wo;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((a.Cout + BN - 1) / BN));
    hipLaunchKernelGGL(kern, grid, dim3(CFG::NT), CFG::LDS_BYTES, st, a);
    return check_launch("conv2d");
}

template <typename T, int BM, int BN, int KP, int NS>
static int launch_conv2(const ConvArgs& a, hipStream_t st) {
    using CFG = ConvCfg2<T, BM, BN, KP, NS>;
    auto kern = conv_igemm2_kernel<CFG, T>;
    static size_t lds_granted[kMaxDevices] = {};                     // per instantiation
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted, "conv2d")) return 1;
    const long long M = (long long)a.N * a.Ho * a.Wo;
    dim3 grid((unsigned)((M + BM - 1) / BM), (unsigned)((a.Cout + BN - 1) / BN));
    hipLaunchKernelGGL(kern, grid, dim3(256), CFG::LDS_BYTES, st, a);
    return check_launch("conv2d");
}

template <typename T, int BN, int NWAVES = 4, int WGM = 2, int DW = 1>
static int launch_conv_halo(const ConvArgs& a, hipStream_t st) {
    using CFG = ConvCfgH<T, BN, NWAVES, WGM, DW>;
    auto kern = conv_halo_kernel<CFG, T>;
    static size_t lds_granted[kMaxDevices] = {};                     // per instantiation
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted, "conv2d")) return 1;
    if (a.stride != 1 || a.shuffle2 || a.korder || a.KH > 3 || a.KW > 3) return set_error("conv2d: the halo tile needs a stride-1 kernel of at most 3x3 taps in K order 0");
    const int tx = (a.W + CFG::PW - 1) / CFG::PW, ty = (a.H + CFG::PH - 1) / CFG::PH;
    dim3 grid((unsigned)(a.N * tx * ty), (unsigned)((a.Cout + BN - 1) / BN));
    hipLaunchKernelGGL(kern, grid, dim3(CFG::NT), CFG::LDS_BYTES, st, a, tx, ty);
    return check_launch("conv2d");
}

template <typename T, int BN, int CH, int PH = 4, int PW = 32>
static int launch_conv_frag(const ConvArgs& a, hipStream_t st) {
    if constexpr (sizeof(T) != 2) {
        return set_error("conv2d: K order 2 (fragment stream) is an fp16 layout");
    } else {
        using CFG = ConvCfgF<T, BN, CH, PH, PW>;
        const bool two = a.epi == S2M2_EPI_GRU || a.epi == S2M2_EPI_GATEMIX;
        const int naux = a.epi == S2M2_EPI_NONE ? 0 : two ? 2 : 1;   // epilogue operands parked in registers
        // (two operands of 8 pieces each do not fit the 128-pixel block's register budget: rejected below, the instantiation is a dummy)
        auto kern = naux == 0 ? conv_frag_kernel<CFG, T, 0> : (naux == 2 && PH == 2) ? conv_frag_kernel<CFG, T, (PH == 2 ? 2 : 1)>
                                                                                      : conv_frag_kernel<CFG, T, 1>;
        static size_t lds_granted[3][kMaxDevices] = {};                 // per instantiation and epilogue-operand variant
        if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted[naux], "conv2d")) return 1;
        if (a.stride != 1 || a.shuffle2 || a.KH > 3 || a.KW > 3 || a.KH * a.KW < 2 || a.Cout % BN || a.Cin % 8 || a.ln_wsum)
            return set_error("conv2d: K order 2 needs a stride-1 3x3 / 3x1 / 1x3 layer with Cout a multiple of %d (Cout=%d)", BN, a.Cout);
        if (a.epi == S2M2_EPI_DUALMIX || (PH == 4 && (a.epi == S2M2_EPI_GRU || a.epi == S2M2_EPI_GATEMIX)))
            return set_error("conv2d: K order 2 with 128-pixel blocks takes one-operand epilogues only (epi=%d has two)", a.epi);
        if (PW != 32 && naux == 2) return set_error("conv2d: K order 2 with 160-pixel blocks takes one-operand epilogues only (epi=%d has two)", a.epi);
        const int tx = (a.W + CFG::PW - 1) / CFG::PW, ty = (a.H + CFG::PH - 1) / CFG::PH;
        dim3 grid((unsigned)(a.N * tx * ty), (unsigned)(a.Cout / BN));
        hipLaunchKernelGGL(kern, grid, dim3(CFG::NT), CFG::LDS_BYTES, st, a, tx, ty);
        return check_launch("conv2d");
    }
}

template <typename T, int BN>
static int launch_conv_pw(const ConvArgs& a, hipStream_t st) {
    using CFG = ConvCfgP<T, BN>;
    auto kern = conv_pw_kernel<CFG, T>;
    if (a.KH != 1 || a.KW != 1 || a.stride != 1) return set_error("conv2d: the pointwise kernel needs a 1x1 stride-1 layer");
    const size_t lds = CFG::lds_bytes(a.Cin);
    if (lds > 160 * 1024) return set_error("conv2d: pointwise kernel: Cin=%d needs %zu bytes of LDS", a.Cin, lds);
    static size_t lds_granted[kMaxDevices] = {};                     // per instantiation
    if (reserve_lds(reinterpret_cast<const void*>(kern), lds, lds_granted, "conv2d")) return 1;
    const long long M = (long long)a.N * a.H * a.W;
    const int ntiles = (int)((M + CFG::BM - 1) / CFG::BM);
    const int per_cu = (int)(160 * 1024 / lds) < 4 ? (int)(160 * 1024 / lds) : 4;      // co-resident blocks per CU
    const int ny = (a.Cout + BN - 1) / BN;
    int gx = 256 * per_cu / ny;                                   // persistent grid: one wave of blocks over the chip
    gx = gx < 1 ? 1 : gx;
    gx = gx > ntiles ? ntiles : gx;
    hipLaunchKernelGGL(kern, dim3(gx, ny), dim3(256), lds, st, a, ntiles);
    return check_launch("conv2d");
}

template <typename T>
static int dispatch_conv(const ConvArgs& a, int tile, hipStream_t st) {
    const long long M = (long long)a.N * a.Ho * a.Wo;
    const bool auto_tile = tile == 0;
    if (a.korder == 2) {                                          // weights packed as a fragment stream: one kernel takes them
        // 64-pixel blocks where 128-pixel blocks would leave most of the 256 CUs without one (tile: 2 / 4 force the patch height)
        static const int force_ph = getenv("S2M2_FRAG_PH") ? atoi(getenv("S2M2_FRAG_PH")) : 0;   // A/B switch
        const long long blocks4 = (long long)a.N * ((a.W + 31) / 32) * ((a.H + 3) / 4) * (a.Cout / 128);
        // layers with an epilogue operand: always 64-pixel blocks (4 operand pieces per thread instead of 8, 160 registers: three blocks per
        // CU -- measured -190 us per pair against the v3 tiles, where the 128-pixel variant was +90 us)
        // (32-pixel blocks, PH = 1, for the 1/8 and 1/16 levels -- twice the blocks again, 8 MFMAs per tap and wave -- measured round 4:
        // 8.18 / 8.18 / 8.21 ms per pair at thresholds 0 / 320 / 640 blocks, same box, 3 alternating runs: no gain, not kept)
        const int ph = tile == 2 || tile == 4 ? tile : (force_ph == 2 || force_ph == 4) ? force_ph
                       : (a.epi != S2M2_EPI_NONE || blocks4 <= 256) ? 2 : 4;
        // 4x40 patches (160 pixels, 5 MFMA tiles) instead of 4x32 where that saves a partial round of blocks: cost = rounds of the 512
        // co-resident block slots (2 per CU) times MFMA tiles per block (tile 40 forces it, S2M2_FRAG_PW=32 switches it off)
        static const int force_pw = getenv("S2M2_FRAG_PW") ? atoi(getenv("S2M2_FRAG_PW")) : 0;         // A/B switch
        bool wide = false;
        static const int aux_pw = getenv("S2M2_FRAG_AUX_PW") ? atoi(getenv("S2M2_FRAG_AUX_PW")) : 40;    // A/B switch: 32 = one-operand layers on 64-pixel blocks only
        const bool one_op = a.epi == S2M2_EPI_ADD || a.epi == S2M2_EPI_MUL;
        if (one_op && (tile == 40 || (tile == 0 && aux_pw == 40 && force_ph == 0))) {
            const long long b4 = (long long)a.N * ((a.W + 31) / 32) * ((a.H + 3) / 4) * (a.Cout / 128);
            const long long b5 = (long long)a.N * ((a.W + 39) / 40) * ((a.H + 3) / 4) * (a.Cout / 128);
            wide = tile == 40 || (b4 > 256 && ((b5 + 511) / 512) * 5 < ((b4 + 511) / 512) * 4 + 4);
        } else if (ph == 4 && a.epi == S2M2_EPI_NONE) {
            const long long blocks5 = (long long)a.N * ((a.W + 39) / 40) * ((a.H + 3) / 4) * (a.Cout / 128);
            const long long cost4 = ((blocks4 + 511) / 512) * 4, cost5 = ((blocks5 + 511) / 512) * 5;
            wide = tile == 40 || force_pw == 40 || (tile == 0 && force_pw != 32 && cost5 < cost4);
        }
        if constexpr (sizeof(T) == 2) {
            // Cout a multiple of 192 but not of 128 with Cin a multiple of 192 (the M model's C = 192 layers): blocks of 192 couts (six
            // waves) on 192-channel chunks (12 k16 steps per tap) -- no padded couts, no half-empty chunk; the weight stream is chunked
            // accordingly by the packers (same rule: s2m2_conv_frag_chunk).  One block per CU (100 KB halo tile): 256 block slots per round
            if (s2m2_conv_frag_chunk(a.Cout, a.Cin) == 192) {
                const long long b4 = (long long)a.N * ((a.W + 31) / 32) * ((a.H + 3) / 4) * (a.Cout / 192);
                const long long b5 = (long long)a.N * ((a.W + 39) / 40) * ((a.H + 3) / 4) * (a.Cout / 192);
                // same choices as below with 256 slots per round: 64-pixel blocks for layers with an epilogue operand and for small grids,
                // 4 x 40 patches where they save a partial round
                const int ph192 = tile == 2 || tile == 4 ? tile : (a.epi != S2M2_EPI_NONE || b4 <= 128) ? 2 : 4;
                const bool one = a.epi == S2M2_EPI_ADD || a.epi == S2M2_EPI_MUL;
                bool wide192 = false;
                if (one && (tile == 40 || tile == 0)) wide192 = tile == 40 || (b4 > 128 && ((b5 + 255) / 256) * 5 < ((b4 + 255) / 256) * 4 + 4);
                else if (ph192 == 4 && a.epi == S2M2_EPI_NONE) wide192 = tile == 40 || (tile == 0 && ((b5 + 255) / 256) * 5 < ((b4 + 255) / 256) * 4);
                if (wide192) return launch_conv_frag<T, 192, 192, 4, 40>(a, st);
                return ph192 == 2 ? launch_conv_frag<T, 192, 192, 2>(a, st) : launch_conv_frag<T, 192, 192, 4>(a, st);
            }
            if (wide) return launch_conv_frag<T, 128, 128, 4, 40>(a, st);
            return ph == 2 ? launch_conv_frag<T, 128, 128, 2>(a, st) : launch_conv_frag<T, 128, 128, 4>(a, st);
        } else {
            return launch_conv_frag<T, 128, 128, 4>(a, st);      // (reports the dtype error)
        }
    }
    if (a.pool2) {                                                // AvgPool2d(2) + 1x1 (the coarse grids): 64x64 tiles, 64- / 128-byte K rows
        if (tile == 2 || (tile != 6 && a.Cin > 512)) return launch_conv<T, 64, 64, 2, 8, 1, 4, 3>(a, st);
        return launch_conv<T, 64, 64, 2, 4, 1, 4, 3>(a, st);
    }
    static const long long t20_min = getenv("S2M2_T20_MIN") ? atoll(getenv("S2M2_T20_MIN")) : 300;   // tuning only
    static const int small_tile = getenv("S2M2_SMALL_TILE") ? atoi(getenv("S2M2_SMALL_TILE")) : 0;
    if (tile == 0) {                                              // measured on MI355X (tools/convbench.py, profiles/r01)
        const int Ktot = a.KH * a.KW * a.Cin;
        if (a.KH * a.KW > 1 && a.KH <= 3 && a.KW <= 3 && a.stride == 1 && !a.shuffle2 && !a.korder && a.Cin > 16) {
            static const bool no8 = getenv("S2M2_CONV_NO_HALO8") != nullptr;    // A/B switch
            static const long long big_min = getenv("S2M2_HALO_BIG_MIN") ? atoll(getenv("S2M2_HALO_BIG_MIN")) : 30000;   // tuning only
            static const int narrow = getenv("S2M2_HALO_NARROW") ? atoi(getenv("S2M2_HALO_NARROW")) : 13;
            static const int coarse = getenv("S2M2_HALO_COARSE") ? atoi(getenv("S2M2_HALO_COARSE")) : 24;
            tile = (a.Cout >= 128 && !no8) ? (M >= big_min ? 26 : coarse) : narrow;   // 8-wave tiles with 2 / 4 weight tiles in flight
        }      // spatial kernels: halo tile; 8 waves x 128 couts when there is enough work
        else if (a.KH * a.KW > 1 && a.Cin <= 16 && a.stride == 1) tile = 6;   // spatial kernel on <= 16 channels: a 128-byte halo chunk would be
                                                                      // mostly padding; K = taps x channels packed densely instead (8->32 full res: 108 vs 156 us)
        else if (a.Cout <= 32) tile = 3;                               // 128x32: narrow heads
        else if (a.Cout >= 128 && ((M + 127) / 128) * ((a.Cout + 127) / 128) >= t20_min) tile = 20;  // 128x128, 64-byte K rows, 8 waves
        else tile = small_tile ? small_tile : (Ktot <= 512 ? 6 : 2);   // 64x64 with 64- / 128-byte K rows
        static const bool deep = getenv("S2M2_CONV_NPF") != nullptr;  // A/B switch: 4 K tiles in flight for the v1 tiles
        if (deep && !a.ln_wsum) tile = tile == 6 ? 16 : tile == 2 ? 17 : tile == 20 ? 27 : tile;
    }
    if (a.ln_wsum) {                                              // pre-LN folded in: the v1 tiles the heuristic picks for 1x1 layers
        switch (tile) {
            case 2: return launch_conv<T, 64, 64, 2, 8, 1, 4, 1>(a, st);
            case 3: return launch_conv<T, 128, 32, 4, 8, 1, 4, 1>(a, st);
            case 6: return launch_conv<T, 64, 64, 2, 4, 1, 4, 1>(a, st);
            case 20: return launch_conv<T, 128, 128, 2, 4, 1, 8, 1>(a, st);
            default: return set_error("conv2d: tile %d has no pre-LayerNorm variant (2, 3, 6, 20 do)", tile);
        }
    }
    if (a.epi == S2M2_EPI_DUALMIX) {                              // two GEMMs, one launch: the v1 tiles the heuristic picks for 1x1 layers
        if (auto_tile) tile = 2;                                  // measured end to end: 64x64 / 128-byte K rows (the 8-wave 128x128 tile needs 168 VGPRs with two accumulator sets: one block per CU)
        if constexpr (sizeof(T) == 2) {
            switch (tile) {
                case 2: return launch_conv<T, 64, 64, 2, 8, 1, 4, 2>(a, st);
                case 6: return launch_conv<T, 64, 64, 2, 4, 1, 4, 2>(a, st);
                case 20: return launch_conv<T, 128, 128, 2, 4, 1, 8, 2>(a, st);
                default: return set_error("conv2d: tile %d has no dual-GEMM variant (2, 6, 20 do)", tile);
            }
        } else {
            switch (tile) {                                       // fp32: 64x64 tiles only (4 staged pieces per thread)
                case 2: case 20: return launch_conv<T, 64, 64, 2, 8, 1, 4, 2>(a, st);
                case 6: return launch_conv<T, 64, 64, 2, 4, 1, 4, 2>(a, st);
                default: return set_error("conv2d: tile %d has no dual-GEMM variant (2, 6, 20 do)", tile);
            }
        }
    }
    switch (tile) {
        case 1: return launch_conv<T, 128, 128, 2>(a, st);
        case 2: return launch_conv<T, 64, 64, 2>(a, st);
        case 3: return launch_conv<T, 128, 32, 4>(a, st);
        case 4: return launch_conv<T, 128, 64, 2>(a, st);
        case 5: return launch_conv<T, 128, 128, 2, 4>(a, st);      // 64-byte K rows: half the LDS, 3 blocks per CU
        case 6: return launch_conv<T, 64, 64, 2, 4>(a, st);
        case 7: return launch_conv2<T, 128, 128, 1, 4>(a, st);     // v2 (LDS-direct ring): 64 KB, 3 tiles ahead
        case 8: return launch_conv2<T, 128, 128, 2, 2>(a, st);     // v2: 128-byte K rows, 1 tile ahead
        case 9: return launch_conv2<T, 128, 128, 2, 3>(a, st);     // v2: 96 KB, 2 tiles ahead
        case 10: return launch_conv2<T, 64, 64, 2, 4>(a, st);      // v2: 64x64, 64 KB
        case 11: return launch_conv2<T, 64, 64, 1, 4>(a, st);      // v2: 64x64, 32 KB
        case 12: return launch_conv_halo<T, 128>(a, st);           // v3 halo tile, 4x32 pixel patch x 128 couts
        case 13: return launch_conv_halo<T, 64>(a, st);            // v3 halo tile, x 64 couts
        case 14: return launch_conv_pw<T, 128>(a, st);             // v4 persistent pointwise, 128 couts per block
        case 15: return launch_conv_pw<T, 64>(a, st);              // v4 persistent pointwise, 64 couts per block
        case 16: return launch_conv<T, 64, 64, 2, 4, 4>(a, st);    // 64x64, 64-byte K rows, 4 K tiles in flight
        case 17: return launch_conv<T, 64, 64, 2, 8, 4>(a, st);    // 64x64, 128-byte K rows, 4 K tiles in flight
        case 18: return launch_conv<T, 128, 128, 2, 4, 4>(a, st);  // 128x128, 64-byte K rows, 4 K tiles in flight
        case 19: return launch_conv_halo<T, 128, 8>(a, st);        // v3 halo tile, 128 couts, 8 waves (32 couts per wave)
        case 20: return launch_conv<T, 128, 128, 2, 4, 1, 8>(a, st);   // 128x128, 64-byte K rows, 8 waves (64 px x 32 couts each)
        case 21: return launch_conv<T, 128, 128, 2, 8, 1, 8>(a, st);   // 128x128, 128-byte K rows, 8 waves
        case 22: return launch_conv<T, 64, 128, 2, 4, 1, 8>(a, st);    // 64x128, 64-byte K rows, 8 waves (32 px x 32 couts each)
        case 23: return launch_conv_halo<T, 64, 8, 4>(a, st);      // v3 halo tile, 64 couts, 8 waves (one patch row x 32 couts each)
        case 27: return launch_conv<T, 128, 128, 2, 4, 4, 8>(a, st);   // t20 with 4 K tiles in flight
        case 24: return launch_conv_halo<T, 64, 8, 4, 4>(a, st);   // t23 with 4 weight tiles in flight (short grids)
        case 25: return launch_conv_halo<T, 64, 4, 2, 4>(a, st);   // t13 with 4 weight tiles in flight
        case 26: return launch_conv_halo<T, 128, 8, 2, 2>(a, st);  // t19 with 2 weight tiles in flight
        default: return set_error("conv2d: unknown tile id %d", tile);
    }
}

}  // namespace s2m2

#if S2M2_FRAG_TRACE
extern "C" int s2m2_debug_frag_trace_clear(void) {
    void* dev = nullptr;
    if (hipGetSymbolAddress(&dev, HIP_SYMBOL(s2m2::g_frag_trace)) != hipSuccess) return 1;
    return hipMemset(dev, 0, sizeof(s2m2::g_frag_trace)) == hipSuccess ? 0 : 1;
}
extern "C" int s2m2_debug_frag_trace(void* host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(s2m2::g_frag_trace), bytes < sizeof(s2m2::g_frag_trace) ? bytes : sizeof(s2m2::g_frag_trace)) == hipSuccess ? 0 : 1;
}
#endif

static int conv2d_impl(const s2m2_conv_desc* d, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(d, "conv2d: null descriptor");
    S2M2_REQUIRE(d->nsrc >= 1 && d->nsrc <= 4, "conv2d: nsrc=%d (1..4)", d->nsrc);
    S2M2_REQUIRE(d->weight && d->out, "conv2d: null weight/out");
    S2M2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && (long long)d->N * d->H * d->W < (1LL << 24), "conv2d: bad shape (at most 2^24 pixels) N=%d H=%d W=%d", d->N, d->H, d->W);
    S2M2_REQUIRE((d->KH & 1) && (d->KW & 1) && d->KH * d->KW <= 32, "conv2d: kernel %dx%d must be odd with at most 32 taps", d->KH, d->KW);
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
            S2M2_REQUIRE((long long)d->N * d->H * d->W * d->src_stride[s] < (1LL << 31), "conv2d: src[%d] has 2^31 or more elements", s);
            a.Cin += d->src_c[s];
        }
    }
    S2M2_REQUIRE(d->act >= S2M2_ACT_NONE && d->act <= S2M2_ACT_TANH, "conv2d: unknown activation %d", d->act);
    S2M2_REQUIRE(d->epi >= S2M2_EPI_NONE && d->epi <= S2M2_EPI_DUALMIX, "conv2d: unknown epilogue %d", d->epi);
    if (d->epi != S2M2_EPI_NONE) {
        S2M2_REQUIRE(d->aux0 && d->aux0_stride % 8 == 0, "conv2d: epilogue %d needs aux0 (stride multiple of 8)", d->epi);
        if (d->epi == S2M2_EPI_GRU || d->epi == S2M2_EPI_GATEMIX || d->epi == S2M2_EPI_DUALMIX)
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
    // epi_cout0: aux0 covers couts >= epi_cout0 only and the kernel indexes it with the cout -- its rows start epi_cout0 elements "before" the
    // tensor (never dereferenced there).  ABI 500: the descriptor carries the tensor's own base (up to 400 the caller passed the shifted
    // pointer: a pointer outside its buffer cannot follow the buffer in a recorded plan, s2m2_plan_end)
    if (d->epi_cout0 > 0 && d->aux0)
        a.aux0 = static_cast<const char*>(d->aux0) - (size_t)d->epi_cout0 * (d->dtype == S2M2_F16 ? 2 : 4);
    a.aux0_stride = d->aux0_stride; a.aux1_stride = d->aux1_stride;
    a.out_scale = d->out_scale; a.shuffle2 = d->shuffle2; a.korder = d->korder;
    a.ln_wsum = d->ln_wsum; a.ln_eps = d->ln_eps;
    a.ksplit = d->ksplit; a.bias2 = d->bias2;
    a.epi_cout0 = d->epi_cout0;
    if (d->epi_cout0)
        S2M2_REQUIRE(d->korder == 2 && d->epi_cout0 > 0 && d->epi_cout0 % s2m2_conv_frag_chunk(d->Cout, a.Cin) == 0 && d->epi_cout0 < d->Cout &&
                     (d->epi == S2M2_EPI_ADD || d->epi == S2M2_EPI_MUL),
                     "conv2d: epi_cout0=%d needs K order 2, a one-operand epilogue (ADD / MUL) and a multiple of the layer's cout block "
                     "(s2m2_conv_frag_chunk: 128 or 192) below Cout", d->epi_cout0);
    if (d->epi == S2M2_EPI_DUALMIX)
        S2M2_REQUIRE(d->KH == 1 && d->KW == 1 && d->stride == 1 && !d->shuffle2 && !d->korder && !d->ln_wsum && d->aux1 &&
                     d->ksplit > 0 && d->ksplit < a.Cin && d->ksplit % 64 == 0 && d->act == S2M2_ACT_SIGMOID && d->out_scale == 1.0f,
                     "conv2d: DUALMIX needs a 1x1 stride-1 layer, act SIGMOID, aux0/aux1, 0 < ksplit < Cin with ksplit a multiple of 64");
    if (d->ln_wsum)
        S2M2_REQUIRE(d->KH == 1 && d->KW == 1 && d->stride == 1 && !d->shuffle2 && !d->korder && d->ln_eps > 0.f &&
                     (d->act == S2M2_ACT_NONE || d->act == S2M2_ACT_GELU),
                     "conv2d: pre-LayerNorm needs a 1x1 stride-1 layer with act NONE or GELU and ln_eps > 0");
    a.stride = d->stride; a.Ho = (d->H + d->stride - 1) / d->stride; a.Wo = (d->W + d->stride - 1) / d->stride;
    a.pool2 = d->pool2;
    if (d->pool2) {
        S2M2_REQUIRE(d->pool2 == 1 && d->KH == 1 && d->KW == 1 && d->stride == 1 && !d->shuffle2 && !d->korder && !d->ln_wsum &&
                     d->epi != S2M2_EPI_DUALMIX && d->H >= 2 && d->W >= 2,
                     "conv2d: pool2 needs a plain 1x1 stride-1 layer on an input of at least 2x2 pixels");
        a.stride = 2; a.Ho = d->H / 2; a.Wo = d->W / 2;           // AvgPool2d(2): floor; rows read pixels (2y, 2x) .. (2y+1, 2x+1)
    }
    S2M2_REQUIRE(d->korder >= 0 && d->korder <= 2, "conv2d: korder=%d (0, 1 or 2)", d->korder);
    S2M2_REQUIRE(d->korder != 1 || a.Cin % (d->dtype == S2M2_F16 ? 32 : 16) == 0, "conv2d: korder 1 needs Cin=%d to be a multiple of 64 bytes of channels", a.Cin);
    S2M2_REQUIRE(d->korder != 2 || d->dtype == S2M2_F16, "conv2d: korder 2 (fragment stream) is an fp16 layout");
    a.zero = zero_page();                                         // (first device call: every argument check is above)
    S2M2_REQUIRE(a.zero, "conv2d: cannot allocate the zero page");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->dtype == S2M2_F16) return dispatch_conv<half_t>(a, d->tile, st);
    if (d->dtype == S2M2_F32) return dispatch_conv<float>(a, d->tile, st);
    return set_error("conv2d: unsupported dtype %d", d->dtype);
}
extern "C" int s2m2_conv2d(const s2m2_conv_desc* d, void* stream) {
    return s2m2::plan_dispatch_desc<s2m2_conv_desc>("s2m2_conv2d", &conv2d_impl, d, stream);
}

