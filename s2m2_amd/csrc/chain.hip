// K9 -- chains of 1x1 layers on token rows, one launch (s2m2_mlp_chain).
//
// The transformer blocks of the model (reference attentions.py:311-321 GlobalAttnBlock, :347-355 BasicAttnBlock) end every
// attention with three row-local layers:   z' = z + proj(o);   out = z' + ffn.2(GELU(ffn.0(LayerNorm(z'))))
// and ConvBlock2D (attentions.py:255-281) has the 1x1 branch  convs_1x.2(ReLU(convs_1x.0(z))).  As separate K5 launches
// each of these layers is a 5-7 us dependent kernel at the coarse levels (1/16, 1/32: launch-latency bound, 100+ of them per
// pair) and a full HBM round trip of the activation at 1/4 resolution.  Here a block owns BM rows for the WHOLE chain:
//   * the row tile lives in LDS, ping-ponging between two [BM][C+pad] buffers (the output of a stage is the A operand of the
//     next; the last stage is staged in place and stored with coalesced 16-byte pieces);
//   * all weights of the chain are ONE stream of [C couts x 64 bytes of K] chunks: D = 4 chunks are always in flight in
//     registers (global -> VGPR -> LDS double buffer), and the stream runs across stage boundaries, so the only exposed
//     memory latency is the first chunk of the first stage;
//   * MFMA roles as in K5 (D[cout][row]: a lane owns a row), NW waves x (BM rows x C/NW couts); long row counts run 64-row
//     tiles, short ones (the 1/16 and 1/32 levels: a few thousand rows on 256 CUs) 32-row tiles so that more CUs take part
//     and the per-wave epilogue (GELU on BM*C/NW/64 values per lane, nothing to overlap with at one block per CU) stays short;
//   * pre-LayerNorm of a stage input = row statistics from the A fragments + rowsum(W) correction (epilogue.h);
//   * residuals: `res` (global rows) is added to the output of stage res_stage with coalesced loads issued before the stage's
//     K loop; `carry` adds the output of stage 0 (still in the other LDS buffer) to the last stage of a 3-stage chain.
// Every intermediate is rounded to the I/O dtype exactly where the separate launches round it (tile in LDS instead of HBM).
#include "common.h"
#include "plan.h"
#include "epilogue.h"
#include <stdlib.h>

namespace s2m2 {

struct ChainArgs {
    const void* x;
    const void* res;
    void* out;
    long long x_stride, res_stride, out_stride, rows;
    const void* w[3];
    const float* b[3];
    const float* wsum[3];
    int act[3];
    int res_stage, carry;
    float ln_eps;
    const void* zero;
    // second output (last stage): LayerNorm with affine of the rows just produced (DispInit's layer_norm folded into the launch
    // that writes feature_tr_4x, so that K1 starts its MFMAs on normalised tokens)
    void* ln_out;
    long long ln_out_stride;
    const float* ln_gamma;
    const float* ln_beta;
    float ln_out_eps;
    // fan-out stages behind the chain: nfan further C -> C layers that ALL read the chain's final rows (still in LDS), layer f writing
    // columns [f*C, (f+1)*C) of fan_out -- the fused Q | K | V projection of the attention that follows (pre-LayerNorm folded in)
    const void* fan_w;                  // packed (nfan*C, C)
    const float* fan_b;                 // (nfan*C) or null
    const float* fan_wsum;              // (nfan*C) row sums of fan_w: LayerNorm (no affine, eps ln_eps) of the final rows folded in; null: none
    void* fan_out;
    long long fan_out_stride;
    int nfan;
    // > 0: row tiles are handed to blocks so that the XCD a block runs on (hardware: block b on XCD b % 8) owns the tiles of ONE eighth of
    // every image -- xcd_tiles consecutive tiles per image and XCD.  The consumer K1 places image row y on XCD y / (h / 8): it then finds the
    // normalised tokens in the L2 of the XCD that wrote them.
    int xcd_tiles;
    int pool_h, pool_w;                 // > 0: x is an (N, pool_h, pool_w, C) image tensor and row m = (n, yo, xo) of the (pool_h/2, pool_w/2) grid is the
                                        //      mean of its pixels (2yo, 2xo) .. (2yo+1, 2xo+1) -- nn.AvgPool2d(2) folded into the tile load
};

template <typename T, int C_, int BM_, int NST_, int NW_, int WP_ = 4>
struct ChainCfg {
    static constexpr int C = C_, BM = BM_, NST = NST_, NW = NW_, NT = 64 * NW_;
    // WP_ = 0: DIRECT form (fp16, weights in MFMA-fragment order, s2m2_chain_desc.weight_frag): a wave's weight fragments go from global
    // memory straight into its MFMA operand registers -- no weight tile in LDS, no block barrier inside a stage's K loop; a "chunk" is one
    // k16 step and a whole stage (C / 16 fragments per 32-cout tile) is in flight while the previous stage computes.
    static constexpr bool DIRECT = WP_ == 0;
    static constexpr int WP = DIRECT ? 2 : WP_;           // 16-byte pieces per weight row and chunk: 4 (64-byte K chunks) or 8 (128-byte)
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int BK = WP * VEC;                   // K elements per chunk
    static constexpr int RS = BK + VEC;                   // weight tile row stride in LDS (80 bytes: conflict-free b128 reads)
    static constexpr int KSTEPS = BK / 16;
    static constexpr int ARS = C + VEC;                   // activation tile row stride (16 bytes of padding)
    static constexpr int CRS = ARS;                       // (name used by stage_tile)
    static constexpr int WM = BM, MT = BM / 32, WN = C / NW, NTL = WN / 32;   // every wave: all BM rows x C / NW couts
    static constexpr int CPS = C / BK;                    // chunks per stage
#ifndef S2M2_CHAIN_DEPTH8
#define S2M2_CHAIN_DEPTH8 1
#endif
    // weight chunks in flight.  Direct form: a whole stage (C / 16 fragments) up to C = 256; at C = 384 half a stage -- 12 waves per block are
    // three per SIMD, 168 registers each: 24 fragments in flight spilled (tools/kernel_resources.py)
    // (C = 512: an eighth -- 16 waves per block are four per SIMD, 128 registers each: 16 fragments in flight spilled 68 - 244 bytes, 8 still 8 - 60)
    static constexpr int D = DIRECT ? (CPS > 24 ? CPS / 8 : CPS > 16 ? CPS / 2 : CPS) : WP == 8 ? 2 : (S2M2_CHAIN_DEPTH8 && CPS % 8 == 0 && sizeof(T) == 2 && C <= 256) ? 8 : 4;
    static constexpr int WROWS = NT / WP;                 // weight rows covered by one pass of the loader threads (WP pieces per row)
    static constexpr int B_IT = DIRECT ? WN / 32 : C / WROWS;   // 16-byte weight pieces per thread and chunk (direct: one per 32-cout tile)
    static constexpr int PPR = C / VEC;                   // 16-byte pieces per activation row
    static constexpr int X_IT = BM * PPR / NT;            // activation pieces per thread
    static constexpr size_t A_BYTES = (size_t)BM * ARS * sizeof(T);
    static constexpr size_t W_BYTES = DIRECT ? 0 : (size_t)C * RS * sizeof(T);
    static constexpr size_t LDS_BYTES = 2 * A_BYTES + 2 * W_BYTES;
    // power-of-two rows of 16-byte pieces: the LayerNorm output reduces a row with lane shuffles inside the store pass; other widths (C = 192,
    // 384: 24 / 48 pieces) take a second pass over the staged tile
    static constexpr bool PPR_POW2 = (PPR & (PPR - 1)) == 0;
    static_assert(C % (DIRECT ? 64 : 128) == 0 && CPS % D == 0 && (BM * PPR) % NT == 0 && BM % 32 == 0 && WN % 32 == 0 && C % WROWS == 0, "unsupported chain tile");
    static_assert(!DIRECT || (sizeof(T) == 2 && D * B_IT <= 24), "direct chain: fp16, at most 24 fragments (96 registers) in flight");
    static_assert(LDS_BYTES <= 160 * 1024, "chain tile does not fit the 160 KB LDS");
};

// the weight stream: chunk j of the chain = columns [ch*BK, ch*BK+BK) of the weight of stage j / CPS
template <typename CFG, typename T>
struct ChainStream {
    raw16_t r[CFG::D][CFG::B_IT];
    const T *w0, *w1, *w2, *wf;
    int off, lrow, pc;

    __device__ __forceinline__ void init(const ChainArgs& p, int tid) {
        w0 = static_cast<const T*>(p.w[0]); w1 = static_cast<const T*>(p.w[1]); w2 = static_cast<const T*>(p.w[2]);
        wf = static_cast<const T*>(p.fan_w);
        lrow = tid / CFG::WP; pc = tid % CFG::WP;
        off = lrow * CFG::C + pc * CFG::VEC;
        // direct form: the fragment of 32-cout tile t = wn * NTL + jn and k16 step ch is 64 lanes x 16 bytes at 16-byte slot (t * C/16 + ch) * 64 + lane
        if (CFG::DIRECT) off = (((tid >> 6) * CFG::NTL * CFG::CPS) * 64 + (tid & 63)) * CFG::VEC;
    }
    __device__ __forceinline__ void fetch(int j, int SLOT) {                  // j is block-uniform; SLOT is static after unrolling
        const int st = j / CFG::CPS, ch = j - st * CFG::CPS;
        // masked telescoping sum instead of a select chain (the compiler folds chains into a scratch lookup table)
        const long long d1 = (const char*)w1 - (const char*)w0, d2 = (const char*)w2 - (const char*)w1;
        const T* wp = reinterpret_cast<const T*>((const char*)w0 + ((st >= 1 ? d1 : 0) + (st >= 2 ? d2 : 0)));
        if (st >= CFG::NST) wp = wf + (size_t)(st - CFG::NST) * CFG::C * CFG::C;      // fan-out stages (block-uniform)
        if constexpr (CFG::DIRECT) {
            const T* q = wp + off + ch * (64 * CFG::VEC);
#pragma unroll
            for (int it = 0; it < CFG::B_IT; ++it) r[SLOT][it] = global_load16(q + it * (CFG::CPS * 64 * CFG::VEC));
        } else {
            const T* q = wp + off + ch * CFG::BK;
#pragma unroll
            for (int it = 0; it < CFG::B_IT; ++it) r[SLOT][it] = global_load16(q + (size_t)it * CFG::WROWS * CFG::C);
        }
    }
    // direct form: k16 step ch of the stage whose fragment-ordered weight starts at `base` (block-uniform)
    __device__ __forceinline__ void fetch_direct(const T* base, int ch, int SLOT) {
        const T* q = base + off + ch * (64 * CFG::VEC);
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it) r[SLOT][it] = global_load16(q + it * (CFG::CPS * 64 * CFG::VEC));
    }
    __device__ __forceinline__ void stash(T* wb, int SLOT) const {
#pragma unroll
        for (int it = 0; it < CFG::B_IT; ++it)
            *reinterpret_cast<raw16_t*>(wb + (size_t)(lrow + CFG::WROWS * it) * CFG::RS + pc * CFG::VEC) = r[SLOT][it];
    }
};

template <typename CFG, typename T, bool LN>
__device__ __forceinline__ void chain_stage_tile(int act, const float16_t (&acc)[CFG::MT][CFG::NTL], T* dst, const CoutRegs<CFG>& bias, int wn,
                                                 int lane, const LnRow* ln, const CoutRegs<CFG>* wsum) {
    switch (act) {                                                // block-uniform
        case S2M2_ACT_GELU: stage_tile<CFG, T, S2M2_ACT_GELU, LN>(acc, dst, bias, 1.0f, 0, wn, lane, ln, wsum); break;
        case S2M2_ACT_RELU: stage_tile<CFG, T, S2M2_ACT_RELU, LN>(acc, dst, bias, 1.0f, 0, wn, lane, ln, wsum); break;
        default: stage_tile<CFG, T, S2M2_ACT_NONE, LN>(acc, dst, bias, 1.0f, 0, wn, lane, ln, wsum); break;
    }
}

template <typename CFG, typename T, int S>
struct ChainStage {
    // one stage: K loop over the CPS chunks of stage S (stream positions S*CPS ...), then the epilogue
    static __device__ __forceinline__ void run(const ChainArgs& p, ChainStream<CFG, T>& ws, T* A0, T* A1, T* W0, T* W1, int tid, long long m0) {
        constexpr int C = CFG::C, BK = CFG::BK, RS = CFG::RS, ARS = CFG::ARS, D = CFG::D, CPS = CFG::CPS, VEC = CFG::VEC;
        const int TOTAL = (CFG::NST + p.nfan) * CPS;               // chunks of the whole weight stream, fan-out stages included
        constexpr bool LAST = S == CFG::NST - 1;
        T* Ain = (S & 1) ? A1 : A0;
        T* Aother = (S & 1) ? A0 : A1;
        T* Aout = LAST ? Ain : Aother;
        const int lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
        const bool ln_on = p.wsum[S] != nullptr;
        const bool res_on = p.res_stage == S;

        // residual rows of this stage: requested now, consumed after the K loop
        raw16_t rr[CFG::X_IT];
        if (res_on) {
#pragma unroll
            for (int it = 0; it < CFG::X_IT; ++it) {
                const int idx = tid + CFG::NT * it, row = idx / CFG::PPR, pcx = idx - row * CFG::PPR;
                const long long m = m0 + row;
                const T* src = m < p.rows ? static_cast<const T*>(p.res) + m * p.res_stride + pcx * VEC : static_cast<const T*>(p.zero);
                rr[it] = global_load16(src);
            }
        }

        CoutRegs<CFG> bias, wsum;                                  // per-cout vectors of this stage: requested now as well
        bias.load(p.b[S], p.zero, C, 0, wn, lane);
        if (ln_on) wsum.load(p.wsum[S], p.zero, C, 0, wn, lane);

        float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        float ln_s[CFG::MT], ln_q[CFG::MT], ln_shift[CFG::MT];
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i) {
            ln_s[i] = ln_q[i] = 0.f;
            ln_shift[i] = (sizeof(T) == 4 && ln_on) ? to_f32(Ain[(size_t)(i * 32 + l31) * ARS]) : 0.f;
        }

        const T* arow = Ain + (size_t)l31 * ARS + hi * 8;
        const int brow = (wn * CFG::WN + l31) * RS + hi * 8;
        if constexpr (CFG::DIRECT) {
            // no LDS weight tile, no barrier: step f multiplies the fragments of ring slot f (requested a whole stage ago), then refills the
            // slot with the same step of the next stage (unconditionally -- after the last stage this stage's fragments are read again and
            // never used: a branch around a load would cost a full `s_waitcnt vmcnt(0)`)
            const T* cur = static_cast<const T*>(p.w[S]);
            const T* nxt = S + 1 < CFG::NST ? static_cast<const T*>(p.w[S + 1 < CFG::NST ? S + 1 : S])
                                            : (p.nfan > 0 ? static_cast<const T*>(p.fan_w) : static_cast<const T*>(p.w[S]));
#pragma unroll
            for (int f = 0; f < CPS; ++f) {                        // ring slot f % D holds step f; it is refilled with the step D ahead in the stream
                Frag<T> xf[CFG::MT], wfr[CFG::NTL];
#pragma unroll
                for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], arow + (size_t)i * 32 * ARS + f * 16);
#pragma unroll
                for (int jn = 0; jn < CFG::NTL; ++jn) wfr[jn].v = __builtin_bit_cast(decltype(wfr[jn].v), ws.r[f % D][jn]);
                if (ln_on) {
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i) ln_accumulate(xf[i], ln_s[i], ln_q[i], ln_shift[i]);
                }
#pragma unroll
                for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                    for (int jn = 0; jn < CFG::NTL; ++jn) mma32(acc[i][jn], wfr[jn], xf[i]);   // D[cout][row]
                if (f + D < CPS) ws.fetch_direct(cur, f + D, f % D);
                else ws.fetch_direct(nxt, f + D - CPS, f % D);
            }
        } else
#pragma unroll 1
        for (int c0 = 0; c0 < CPS; c0 += D) {
#pragma unroll
            for (int f = 0; f < D; ++f) {
                const int j = S * CPS + c0 + f;                   // stream position of this chunk
                T* wb = (f & 1) ? W1 : W0;                        // CPS and D are even: the LDS buffer of chunk j is j & 1 = f & 1
                T* wnext = (f & 1) ? W0 : W1;
                if (j + D < TOTAL) ws.fetch(j + D, f);             // slot f was stashed one chunk ago: refill
                const T* a = arow + (c0 + f) * BK;
                const T* b = wb + brow;
#pragma unroll
                for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
                    Frag<T> xf[CFG::MT], wf[CFG::NTL];
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], a + (size_t)i * 32 * ARS + kk * 16);
#pragma unroll
                    for (int jn = 0; jn < CFG::NTL; ++jn) load_frag(wf[jn], b + (size_t)jn * 32 * RS + kk * 16);
                    if (ln_on) {
#pragma unroll
                        for (int i = 0; i < CFG::MT; ++i) ln_accumulate(xf[i], ln_s[i], ln_q[i], ln_shift[i]);
                    }
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                        for (int jn = 0; jn < CFG::NTL; ++jn) mma32(acc[i][jn], wf[jn], xf[i]);   // D[cout][row]
                }
                if (j + 1 < TOTAL) ws.stash(wnext, (f + 1) % D);
                __syncthreads();
            }
        }

        // direct form: no barrier closed the K loop, and the last stage stages its tile IN PLACE (Aout == Ain): every wave must have read
        // its last A fragment before the first one writes
        if constexpr (CFG::DIRECT && LAST) __syncthreads();
        // ---- epilogue: bias / folded LayerNorm / activation in registers -> Aout[row][cout]
        if (ln_on) {
            LnRow ln[CFG::MT];
            const float inv = 1.0f / (float)C;
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i) {
                const float s = ln_s[i] + __shfl_xor(ln_s[i], 32), q = ln_q[i] + __shfl_xor(ln_q[i], 32);
                const float mean = s * inv;
                ln[i].mean = mean + ln_shift[i];
                ln[i].rstd = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, q * inv), 0.f) + p.ln_eps);
            }
            chain_stage_tile<CFG, T, true>(p.act[S], acc, Aout, bias, wn, lane, ln, &wsum);
        } else {
            chain_stage_tile<CFG, T, false>(p.act[S], acc, Aout, bias, wn, lane, nullptr, nullptr);
        }
        __syncthreads();

        if constexpr (!LAST) {
            if (res_on) {                                          // Aout += res (rounded like the separate launch: tile, then the sum)
#pragma unroll
                for (int it = 0; it < CFG::X_IT; ++it) {
                    const int idx = tid + CFG::NT * it, row = idx / CFG::PPR, pcx = idx - row * CFG::PPR;
                    Vec16<T>* q = reinterpret_cast<Vec16<T>*>(Aout + (size_t)row * ARS + pcx * VEC);
                    Vec16<T> v = *q;
                    const Vec16<T> u = __builtin_bit_cast(Vec16<T>, rr[it]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v.v[e] = from_f32<T>(to_f32(v.v[e]) + to_f32(u.v[e]));
                    *q = v;
                }
                __syncthreads();
            }
        } else {                                                   // coalesced store of the staged tile (+ carry, + res)
            T* outp = static_cast<T*>(p.out);
            // LayerNorm output: a row's PPR pieces sit in PPR consecutive lanes of one wave (NT % PPR == 0, PPR in {16, 32, 64}:
            // checked on the host), and a thread's piece column is the same for every `it`
            const bool ln2 = p.ln_out != nullptr;
            const bool ln2_rows = ln2 && CFG::PPR_POW2;              // statistics by lane shuffles inside this pass (PPR lanes share a row)
            float g2[VEC], b2[VEC];
            if (ln2_rows) {
                const int pcx = tid % CFG::PPR;
#pragma unroll
                for (int e = 0; e < VEC; ++e) { g2[e] = p.ln_gamma[pcx * VEC + e]; b2[e] = p.ln_beta[pcx * VEC + e]; }
            }
#pragma unroll
            for (int it = 0; it < CFG::X_IT; ++it) {
                const int idx = tid + CFG::NT * it, row = idx / CFG::PPR, pcx = idx - row * CFG::PPR;
                const long long m = m0 + row;
                Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(Aout + (size_t)row * ARS + pcx * VEC);
                if (p.carry) {
                    const Vec16<T> u = *reinterpret_cast<const Vec16<T>*>(Aother + (size_t)row * ARS + pcx * VEC);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v.v[e] = from_f32<T>(to_f32(v.v[e]) + to_f32(u.v[e]));
                }
                if (res_on) {
                    const Vec16<T> u = __builtin_bit_cast(Vec16<T>, rr[it]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v.v[e] = from_f32<T>(to_f32(v.v[e]) + to_f32(u.v[e]));
                }
                if (m < p.rows) *reinterpret_cast<Vec16<T>*>(outp + m * p.out_stride + pcx * VEC) = v;
                // the stored rows go back into the tile: A operand of the fan-out stages / input of the second LayerNorm pass
                if (p.nfan > 0 || (ln2 && !CFG::PPR_POW2)) *reinterpret_cast<Vec16<T>*>(Aout + (size_t)row * ARS + pcx * VEC) = v;
                if constexpr (CFG::PPR_POW2) {
                if (ln2_rows) {                                    // block-uniform
                    // statistics of the STORED (rounded) row, two passes in fp32 like K1 / nn.LayerNorm: biased variance, eps inside
                    float x[VEC], sum = 0.f;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { x[e] = to_f32(v.v[e]); sum += x[e]; }
#pragma unroll
                    for (int o = 1; o < CFG::PPR; o <<= 1) sum += __shfl_xor(sum, o);
                    const float mean = sum * (1.0f / (float)C);
                    float sq = 0.f;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { x[e] -= mean; sq = __builtin_fmaf(x[e], x[e], sq); }
#pragma unroll
                    for (int o = 1; o < CFG::PPR; o <<= 1) sq += __shfl_xor(sq, o);
                    const float rstd = rsqrtf(sq * (1.0f / (float)C) + p.ln_out_eps);
                    Vec16<T> o;
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>(__builtin_fmaf(x[e] * rstd, g2[e], b2[e]));
                    if (m < p.rows) {
                        *reinterpret_cast<Vec16<T>*>(static_cast<T*>(p.ln_out) + m * p.ln_out_stride + pcx * VEC) = o;
                    }
                }
                }
            }
            if constexpr (!CFG::PPR_POW2) {
                if (ln2) {                                         // block-uniform
                    // rows of 24 / 48 pieces do not line up with the lanes of the store pass: second pass over the stored tile, 8 lanes per row
                    // (piece q of a row on lane q % 8 -- 128 contiguous bytes per 8 lanes in every read and store), two-pass fp32 statistics
                    // with the 8-lane DPP sum, as K1's own LayerNorm takes them
                    __syncthreads();
                    constexpr int PPL = CFG::PPR / 8;               // pieces per lane
                    static_assert(CFG::PPR % 8 == 0, "a row is a whole number of 8-piece rounds");
                    const int sub = tid & 7;
                    for (int row = tid >> 3; row < CFG::BM; row += CFG::NT / 8) {
                        const long long m = m0 + row;
                        float x[PPL][VEC], sum = 0.f;
#pragma unroll
                        for (int q = 0; q < PPL; ++q) {
                            const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(Aout + (size_t)row * ARS + (sub + 8 * q) * VEC);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) { x[q][e] = to_f32(v.v[e]); sum += x[q][e]; }
                        }
                        const float mean = group8_sum(sum) * (1.0f / (float)C);
                        float sq = 0.f;
#pragma unroll
                        for (int q = 0; q < PPL; ++q)
#pragma unroll
                            for (int e = 0; e < VEC; ++e) { x[q][e] -= mean; sq = __builtin_fmaf(x[q][e], x[q][e], sq); }
                        const float rstd = rsqrtf(group8_sum(sq) * (1.0f / (float)C) + p.ln_out_eps);
#pragma unroll
                        for (int q = 0; q < PPL; ++q) {
                            const int c0 = (sub + 8 * q) * VEC;
                            Vec16<T> o;
#pragma unroll
                            for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>(__builtin_fmaf(x[q][e] * rstd, p.ln_gamma[c0 + e], p.ln_beta[c0 + e]));
                            if (m < p.rows) *reinterpret_cast<Vec16<T>*>(static_cast<T*>(p.ln_out) + m * p.ln_out_stride + c0) = o;
                        }
                    }
                }
            }
        }
    }
};

// Fan-out stages: nfan C -> C layers that all read the chain's final rows (Afin: the last stage's tile with carry / residual added and
// rounded, written back by its store pass), each staged through Aoth and stored to its own column block of fan_out.  The weight stream
// simply continues (stream positions (NST + f) * CPS ...).  Pre-LayerNorm fold as in the regular stages; the row statistics are taken from
// the A fragments during the first fan-out stage and reused.
template <typename CFG, typename T>
__device__ __forceinline__ void chain_fan(const ChainArgs& p, ChainStream<CFG, T>& ws, T* Afin, T* Aoth, T* W0, T* W1, int tid, long long m0) {
    constexpr int C = CFG::C, BK = CFG::BK, RS = CFG::RS, ARS = CFG::ARS, D = CFG::D, CPS = CFG::CPS, VEC = CFG::VEC;
    const int TOTAL = (CFG::NST + p.nfan) * CPS;
    const int lane = tid & 63, wn = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const bool ln_on = p.fan_wsum != nullptr;
    LnRow ln[CFG::MT];
    const T* arow = Afin + (size_t)l31 * ARS + hi * 8;
    const int brow = (wn * CFG::WN + l31) * RS + hi * 8;
    __syncthreads();                                               // the final rows are in Afin (written by every thread of the last store pass)
#pragma unroll 1
    for (int F = 0; F < p.nfan; ++F) {
        CoutRegs<CFG> bias, wsum;
        bias.load(p.fan_b ? p.fan_b + F * C : nullptr, p.zero, C, 0, wn, lane);
        if (ln_on) wsum.load(p.fan_wsum + F * C, p.zero, C, 0, wn, lane);
        float16_t acc[CFG::MT][CFG::NTL];
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
            for (int j = 0; j < CFG::NTL; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        float ln_s[CFG::MT], ln_q[CFG::MT], ln_shift[CFG::MT];
#pragma unroll
        for (int i = 0; i < CFG::MT; ++i) {
            ln_s[i] = ln_q[i] = 0.f;
            ln_shift[i] = (sizeof(T) == 4 && ln_on) ? to_f32(Afin[(size_t)(i * 32 + l31) * ARS]) : 0.f;
        }
        const bool stats = ln_on && F == 0;
        if constexpr (CFG::DIRECT) {
            const T* cur = static_cast<const T*>(p.fan_w) + (size_t)F * C * C;
            const T* nxt = static_cast<const T*>(p.fan_w) + (size_t)(F + 1 < p.nfan ? F + 1 : F) * C * C;
#pragma unroll
            for (int f = 0; f < CPS; ++f) {
                Frag<T> xf[CFG::MT], wfr[CFG::NTL];
#pragma unroll
                for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], arow + (size_t)i * 32 * ARS + f * 16);
#pragma unroll
                for (int jn = 0; jn < CFG::NTL; ++jn) wfr[jn].v = __builtin_bit_cast(decltype(wfr[jn].v), ws.r[f % D][jn]);
                if (stats) {
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i) ln_accumulate(xf[i], ln_s[i], ln_q[i], ln_shift[i]);
                }
#pragma unroll
                for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                    for (int jn = 0; jn < CFG::NTL; ++jn) mma32(acc[i][jn], wfr[jn], xf[i]);
                if (f + D < CPS) ws.fetch_direct(cur, f + D, f % D);
                else ws.fetch_direct(nxt, f + D - CPS, f % D);
            }
        } else
#pragma unroll 1
        for (int c0 = 0; c0 < CPS; c0 += D) {
#pragma unroll
            for (int f = 0; f < D; ++f) {
                const int j = (CFG::NST + F) * CPS + c0 + f;       // stream position of this chunk
                T* wb = (f & 1) ? W1 : W0;
                T* wnext = (f & 1) ? W0 : W1;
                if (j + D < TOTAL) ws.fetch(j + D, f);
                const T* a = arow + (c0 + f) * BK;
                const T* b = wb + brow;
#pragma unroll
                for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
                    Frag<T> xf[CFG::MT], wf[CFG::NTL];
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i) load_frag(xf[i], a + (size_t)i * 32 * ARS + kk * 16);
#pragma unroll
                    for (int jn = 0; jn < CFG::NTL; ++jn) load_frag(wf[jn], b + (size_t)jn * 32 * RS + kk * 16);
                    if (stats) {
#pragma unroll
                        for (int i = 0; i < CFG::MT; ++i) ln_accumulate(xf[i], ln_s[i], ln_q[i], ln_shift[i]);
                    }
#pragma unroll
                    for (int i = 0; i < CFG::MT; ++i)
#pragma unroll
                        for (int jn = 0; jn < CFG::NTL; ++jn) mma32(acc[i][jn], wf[jn], xf[i]);
                }
                if (j + 1 < TOTAL) ws.stash(wnext, (f + 1) % D);
                __syncthreads();
            }
        }
        if (stats) {
            const float inv = 1.0f / (float)C;
#pragma unroll
            for (int i = 0; i < CFG::MT; ++i) {
                const float sm = ln_s[i] + __shfl_xor(ln_s[i], 32), q = ln_q[i] + __shfl_xor(ln_q[i], 32);
                const float mean = sm * inv;
                ln[i].mean = mean + ln_shift[i];
                ln[i].rstd = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, q * inv), 0.f) + p.ln_eps);
            }
        }
        if (ln_on) chain_stage_tile<CFG, T, true>(S2M2_ACT_NONE, acc, Aoth, bias, wn, lane, ln, &wsum);
        else chain_stage_tile<CFG, T, false>(S2M2_ACT_NONE, acc, Aoth, bias, wn, lane, nullptr, nullptr);
        __syncthreads();
        T* outp = static_cast<T*>(p.fan_out) + (size_t)F * C;
#pragma unroll
        for (int it = 0; it < CFG::X_IT; ++it) {
            const int idx = tid + CFG::NT * it, row = idx / CFG::PPR, pcx = idx - row * CFG::PPR;
            const long long m = m0 + row;
            const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(Aoth + (size_t)row * ARS + pcx * VEC);
            if (m < p.rows) *reinterpret_cast<Vec16<T>*>(outp + m * p.fan_out_stride + pcx * VEC) = v;
        }
        __syncthreads();                                           // Aoth is staged again by the next fan-out stage
    }
}

// second launch bound: waves per SIMD asked of the register allocator for the direct form at C = 128.  With the plain bound it spent 182 + 32
// registers on the three-stage 64-row tile (two waves per SIMD); asked for three it fits 168 (14 spilled in the three-stage form, none in the
// others): measured -1.15 % per pair same-box (profiles/r04/ab_minwaves.txt: 8.756 vs 8.858 ms).  S2M2_CHAIN_MINWAVES=1: the round-3 build
#ifndef S2M2_CHAIN_MINWAVES
#define S2M2_CHAIN_MINWAVES 3
#endif
template <typename CFG, typename T>
__global__ __launch_bounds__(CFG::NT, (CFG::DIRECT && CFG::C == 128) ? S2M2_CHAIN_MINWAVES : 1) void mlp_chain_kernel(ChainArgs p) {
    constexpr int VEC = CFG::VEC, ARS = CFG::ARS, D = CFG::D;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* A0 = reinterpret_cast<T*>(smem);
    T* A1 = reinterpret_cast<T*>(smem + CFG::A_BYTES);
    T* W0 = reinterpret_cast<T*>(smem + 2 * CFG::A_BYTES);
    T* W1 = reinterpret_cast<T*>(smem + 2 * CFG::A_BYTES + CFG::W_BYTES);
    const int tid = threadIdx.x;
    int tile = blockIdx.x;
    if (p.xcd_tiles > 0) {                                        // (the host checked that the grid is images x 8 x xcd_tiles)
        const int x = tile & 7, k = tile >> 3;
        const int img = k / p.xcd_tiles, r = k - img * p.xcd_tiles;
        tile = (img * 8 + x) * p.xcd_tiles + r;
    }
    const long long m0 = (long long)tile * CFG::BM;

    ChainStream<CFG, T> ws;
    ws.init(p, tid);
    // the first D chunks of the weight stream and the row tile are requested together
    // (NST == 0: fan-out only -- the stream starts with the first fan-out layer)
    const T* wfirst = static_cast<const T*>(CFG::NST > 0 ? p.w[0] : p.fan_w);
    if constexpr (CFG::DIRECT) ws.fetch_direct(wfirst, 0, 0);
    else ws.fetch(0, 0);
    raw16_t xr[CFG::X_IT];
#pragma unroll
    for (int it = 0; it < CFG::X_IT; ++it) {
        const int idx = tid + CFG::NT * it, row = idx / CFG::PPR, pcx = idx - row * CFG::PPR;
        const long long m = m0 + row;
        if (p.pool_w == 0) {                                       // block-uniform
            const T* src = m < p.rows ? static_cast<const T*>(p.x) + m * p.x_stride + pcx * VEC : static_cast<const T*>(p.zero);
            xr[it] = global_load16(src);
        } else {
            // AvgPool2d(2) in front of the chain (the down_conv of Unet / MRT, unet.py:24-29): the mean of four pixels, formed and rounded
            // exactly as K7's pooling kernel and K5's pool2 operand load do ((a + b + c + d) * 0.25 in fp32, one rounding)
            const int Wo = p.pool_w >> 1, Ho = p.pool_h >> 1;
            const bool ok = m < p.rows;
            const long long mm = ok ? m : 0;
            const int xo = (int)(mm % Wo);
            const long long t = mm / Wo;
            const int yo = (int)(t % Ho);
            const long long n = t / Ho;
            const T* q = static_cast<const T*>(p.x) + ((n * p.pool_h + 2 * yo) * p.pool_w + 2 * xo) * p.x_stride + pcx * VEC;
            const long long dx = p.x_stride, dy = (long long)p.pool_w * p.x_stride;
            const Vec16<T> a = __builtin_bit_cast(Vec16<T>, global_load16(q)), b = __builtin_bit_cast(Vec16<T>, global_load16(q + dx));
            const Vec16<T> c = __builtin_bit_cast(Vec16<T>, global_load16(q + dy)), d = __builtin_bit_cast(Vec16<T>, global_load16(q + dy + dx));
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e)
                o.v[e] = ok ? from_f32<T>((to_f32(a.v[e]) + to_f32(b.v[e]) + to_f32(c.v[e]) + to_f32(d.v[e])) * 0.25f) : from_f32<T>(0.f);
            xr[it] = __builtin_bit_cast(raw16_t, o);
        }
    }
#pragma unroll
    for (int f = 1; f < D; ++f)
        if constexpr (CFG::DIRECT) ws.fetch_direct(wfirst, f, f);     // (D = CPS: the whole first stage)
        else if (f < (CFG::NST + p.nfan) * CFG::CPS) ws.fetch(f, f);
    if constexpr (!CFG::DIRECT) ws.stash(W0, 0);
#pragma unroll
    for (int it = 0; it < CFG::X_IT; ++it) {
        const int idx = tid + CFG::NT * it, row = idx / CFG::PPR, pcx = idx - row * CFG::PPR;
        *reinterpret_cast<raw16_t*>(A0 + (size_t)row * ARS + pcx * VEC) = xr[it];
    }
    __syncthreads();

    if constexpr (CFG::NST > 0) ChainStage<CFG, T, 0>::run(p, ws, A0, A1, W0, W1, tid, m0);
    if constexpr (CFG::NST > 1) ChainStage<CFG, T, 1>::run(p, ws, A0, A1, W0, W1, tid, m0);
    if constexpr (CFG::NST > 2) ChainStage<CFG, T, 2>::run(p, ws, A0, A1, W0, W1, tid, m0);
    if (p.nfan > 0) {                                              // block-uniform
        T* Afin = (CFG::NST > 0 && ((CFG::NST - 1) & 1)) ? A1 : A0;   // the last stage stages in place: its tile is its input buffer (NST == 0: the x rows)
        chain_fan<CFG, T>(p, ws, Afin, Afin == A0 ? A1 : A0, W0, W1, tid, m0);
    }
}


template <typename T, int C, int BM, int NST, int NW, int WP = 4>
static int launch_chain(const ChainArgs& a, hipStream_t st) {
    using CFG = ChainCfg<T, C, BM, NST, NW, WP>;
    auto kern = mlp_chain_kernel<CFG, T>;
    static size_t lds_granted[kMaxDevices] = {};                     // per instantiation
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted, "mlp_chain")) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)((a.rows + BM - 1) / BM)), dim3(CFG::NT), CFG::LDS_BYTES, st, a);
    return check_launch("mlp_chain");
}

template <typename T, int C, int BM, int NW, int WP = 4>
static int launch_chain_n(const ChainArgs& a, int nst, hipStream_t st) {
    if (nst == 1) return launch_chain<T, C, BM, 1, NW, WP>(a, st);
    if (nst == 2) return launch_chain<T, C, BM, 2, NW, WP>(a, st);
    return launch_chain<T, C, BM, 3, NW, WP>(a, st);
}

}  // namespace s2m2

extern "C" int s2m2_mlp_chain_supported(int C, int dtype) {
    if (dtype == S2M2_F16) return C == 128 || C == 256 || C == 384 || C == 512;
    if (dtype == S2M2_F32) return C == 128 || C == 256;
    return 0;
}

// direct form: one wave per 32 couts (C / 32 waves per block: 4, 6, 8, 12, 16), a whole stage of C / 16 fragments in flight per wave (half a stage at
// C = 384 / 512).  C = 512 (r06: the L model's 1/16 and 1/32 levels, 54 launches per forward on the LDS-staged form before): 16 waves = four per SIMD,
// 128 registers each, 32-row tiles only
extern "C" int s2m2_mlp_chain_frag_supported(int C, int dtype) {
    static const bool no512 = getenv("S2M2_CHAIN_FRAG512") != nullptr && atoi(getenv("S2M2_CHAIN_FRAG512")) == 0;      // A/B switch
    return dtype == S2M2_F16 && (C == 128 || C == 192 || C == 256 || C == 384 || (C == 512 && !no512));
}

namespace s2m2 {
// direct form, by width: 32-row tiles while they fit the chip in about one round, else 64-row tiles (half the weight traffic per row)
template <int NST>
static int launch_chain_direct(const ChainArgs& a, int C, bool tall, hipStream_t st) {
    switch (C) {
        case 128: return tall ? launch_chain<half_t, 128, 64, NST, 4, 0>(a, st) : launch_chain<half_t, 128, 32, NST, 4, 0>(a, st);
        case 192: return tall ? launch_chain<half_t, 192, 64, NST, 6, 0>(a, st) : launch_chain<half_t, 192, 32, NST, 6, 0>(a, st);
        case 256: return tall ? launch_chain<half_t, 256, 64, NST, 8, 0>(a, st) : launch_chain<half_t, 256, 32, NST, 8, 0>(a, st);
        case 512: return launch_chain<half_t, 512, 32, NST, 16, 0>(a, st);
        default: return tall ? launch_chain<half_t, 384, 64, NST, 12, 0>(a, st) : launch_chain<half_t, 384, 32, NST, 12, 0>(a, st);
    }
}
static bool chain_direct_tall(int C, long long rows) {
    static const int force_bm = getenv("S2M2_CHAIN_DIRECT_BM") ? atoi(getenv("S2M2_CHAIN_DIRECT_BM")) : 0;      // 32 / 64 forces one (tuning)
    if (C >= 384) return false;                                    // (64-row tiles at 12 waves per block spill: 55 - 88 registers; 16 waves: 128 registers each)
    return force_bm ? force_bm == 64 : rows > (C == 128 ? 24576 : C == 192 ? 16384 : 8192);
}
}  // namespace s2m2

extern "C" int s2m2_mlp_fan_supported(int C, int nfan, int dtype) { return s2m2_mlp_chain_frag_supported(C, dtype) && nfan >= 1 && nfan <= 4; }

static int mlp_chain_impl(const s2m2_chain_desc* d, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(d, "mlp_chain: null descriptor");
    S2M2_REQUIRE(d->x, "mlp_chain: null x");
    S2M2_REQUIRE(d->weight_frag == 0 || (d->weight_frag == 1 && (d->nstage > 0 || d->nfan > 0) && s2m2_mlp_chain_frag_supported(d->C, d->dtype)),
                 "mlp_chain: weight_frag=%d needs fp16 and C = 128 / 192 / 256 / 384 / 512", d->weight_frag);
    S2M2_REQUIRE((d->pool_h == 0 && d->pool_w == 0) ||
                 (d->weight_frag && d->pool_h >= 2 && d->pool_w >= 2 && d->rows % ((long long)(d->pool_h / 2) * (d->pool_w / 2)) == 0),
                 "mlp_chain: pool_h / pool_w need weight_frag, an input of at least 2x2 pixels and rows = N * (pool_h/2) * (pool_w/2)");
    S2M2_REQUIRE((d->pool_h == 0 && d->pool_w == 0) || (d->res_stage < 0 && !d->carry && !d->ln_out && d->xcd_group_rows == 0),
                 "mlp_chain: no residual, carry, ln_out or xcd_group_rows with a pooled tile load (pool_h / pool_w)");
    if (d->nstage == 0 && d->weight_frag) {
        // fan-out only, direct form: the nfan layers read the x rows, their fragments straight from global memory (any row count)
        S2M2_REQUIRE(d->nfan >= 1 && d->nfan <= 4, "mlp_chain: nfan=%d (1..4)", d->nfan);
        S2M2_REQUIRE(d->rows > 0 && d->rows < (1LL << 31) && d->x_stride >= d->C && d->x_stride % 8 == 0, "mlp_chain: bad rows / x_stride");
        S2M2_REQUIRE(d->fan_weight && d->fan_out && d->fan_out_stride >= (long long)d->nfan * d->C && d->fan_out_stride % 8 == 0,
                     "mlp_chain: fan-out stages need fan_weight, fan_out and a row stride of at least nfan * C (multiple of 8)");
        S2M2_REQUIRE(!d->fan_ln_wsum || d->ln_eps > 0.f, "mlp_chain: ln_eps must be positive");
        ChainArgs f{};
        f.x = d->x; f.x_stride = d->x_stride; f.rows = d->rows; f.ln_eps = d->ln_eps; f.res_stage = -1;
        f.fan_w = d->fan_weight; f.fan_b = d->fan_bias; f.fan_wsum = d->fan_ln_wsum; f.fan_out = d->fan_out; f.fan_out_stride = d->fan_out_stride;
        f.nfan = d->nfan;
        f.pool_h = d->pool_h; f.pool_w = d->pool_w;
        f.zero = zero_page();
        S2M2_REQUIRE(f.zero, "mlp_chain: cannot allocate the zero page");
        hipStream_t fst = static_cast<hipStream_t>(stream);
        return launch_chain_direct<0>(f, d->C, chain_direct_tall(d->C, d->rows), fst);
    }
    S2M2_REQUIRE(d->nstage != 0, "mlp_chain: fan-out only (nstage = 0) exists in the direct form: weight_frag = 1, fp16, C = 128 / 192 / 256 / 384 / 512");
    S2M2_REQUIRE(d->out, "mlp_chain: null out");
    S2M2_REQUIRE(d->nstage >= 1 && d->nstage <= 3, "mlp_chain: nstage=%d (0..3)", d->nstage);
    S2M2_REQUIRE(d->weight_frag || s2m2_mlp_chain_supported(d->C, d->dtype), "mlp_chain: C=%d dtype=%d is not supported (fp16: 128/256/384/512, fp32: 128/256; direct form fp16: 128/192/256/384/512)", d->C, d->dtype);
    S2M2_REQUIRE(d->rows > 0 && d->rows < (1LL << 31), "mlp_chain: rows=%lld", d->rows);
    S2M2_REQUIRE(d->x_stride >= d->C && d->x_stride % 8 == 0 && d->out_stride >= d->C && d->out_stride % 8 == 0,
                 "mlp_chain: row strides %lld/%lld must be multiples of 8 and at least C", d->x_stride, d->out_stride);
    S2M2_REQUIRE(d->res_stage >= -1 && d->res_stage < d->nstage, "mlp_chain: res_stage=%d", d->res_stage);
    if (d->res_stage >= 0) S2M2_REQUIRE(d->res && d->res_stride >= d->C && d->res_stride % 8 == 0, "mlp_chain: res_stage needs res rows (stride multiple of 8)");
    S2M2_REQUIRE(!d->carry || d->nstage == 3, "mlp_chain: carry (output of stage 0 added to the last stage) needs 3 stages");
    ChainArgs a;
    a.x = d->x; a.res = d->res; a.out = d->out;
    a.x_stride = d->x_stride; a.res_stride = d->res_stride; a.out_stride = d->out_stride; a.rows = d->rows;
    bool any_ln = false;
    for (int s = 0; s < 3; ++s) {
        const bool on = s < d->nstage;
        a.w[s] = on ? d->weight[s] : nullptr; a.b[s] = on ? d->bias[s] : nullptr; a.wsum[s] = on ? d->ln_wsum[s] : nullptr;
        a.act[s] = on ? d->act[s] : 0;
        if (on) {
            S2M2_REQUIRE(d->weight[s], "mlp_chain: weight[%d] is null", s);
            S2M2_REQUIRE(d->act[s] == S2M2_ACT_NONE || d->act[s] == S2M2_ACT_GELU || d->act[s] == S2M2_ACT_RELU,
                         "mlp_chain: act[%d]=%d (NONE, GELU or RELU)", s, d->act[s]);
            any_ln |= d->ln_wsum[s] != nullptr;
        }
    }
    S2M2_REQUIRE(!any_ln || d->ln_eps > 0.f, "mlp_chain: ln_eps must be positive");
    a.res_stage = d->res_stage; a.carry = d->carry; a.ln_eps = d->ln_eps;
    a.ln_out = d->ln_out; a.ln_out_stride = d->ln_out_stride; a.ln_gamma = d->ln_gamma; a.ln_beta = d->ln_beta; a.ln_out_eps = d->ln_out_eps;
    a.pool_h = d->pool_h; a.pool_w = d->pool_w;
    a.fan_w = d->fan_weight; a.fan_b = d->fan_bias; a.fan_wsum = d->fan_ln_wsum; a.fan_out = d->fan_out; a.fan_out_stride = d->fan_out_stride;
    a.nfan = d->nfan;
    S2M2_REQUIRE(d->nfan >= 0 && d->nfan <= 4, "mlp_chain: nfan=%d (0..4)", d->nfan);
    if (d->nfan > 0) {
        S2M2_REQUIRE(d->fan_weight && d->fan_out && d->fan_out_stride >= (long long)d->nfan * d->C && d->fan_out_stride % 8 == 0,
                     "mlp_chain: fan-out stages need fan_weight, fan_out and a row stride of at least nfan * C (multiple of 8)");
        S2M2_REQUIRE(!d->fan_ln_wsum || d->ln_eps > 0.f, "mlp_chain: ln_eps must be positive");
    }
    if (d->ln_out) {
        const int ppr = d->C * (d->dtype == S2M2_F16 ? 2 : 4) / 16;      // 16-byte pieces per row = lanes that share a row in the store pass
        S2M2_REQUIRE(ppr % 8 == 0 && ppr <= 64, "mlp_chain: ln_out needs a row of 8 k <= 64 16-byte pieces (C=%d has %d)", d->C, ppr);
        S2M2_REQUIRE(d->ln_gamma && d->ln_beta && d->ln_out_eps > 0.f && d->ln_out_stride >= d->C && d->ln_out_stride % 8 == 0,
                     "mlp_chain: ln_out needs gamma, beta, a positive eps and a row stride that is a multiple of 8");
    }
    a.zero = zero_page();
    S2M2_REQUIRE(a.zero, "mlp_chain: cannot allocate the zero page");
    a.xcd_tiles = 0;
    static const bool xcd_off = getenv("S2M2_K9_XCD") != nullptr && atoi(getenv("S2M2_K9_XCD")) == 0;    // A/B switch
    const int bm = (d->dtype == S2M2_F32 || d->rows <= 8192 || d->C >= 384) ? 32 : 64;   // (the tile heights picked below)
    if (d->xcd_group_rows > 0 && !xcd_off && d->xcd_group_rows % bm == 0 && d->rows % (8LL * d->xcd_group_rows) == 0)
        a.xcd_tiles = (int)(d->xcd_group_rows / bm);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const char cfg = d->rows <= 8192 ? 's' : 'm';                 // at most one 32-row tile per CU: short tiles, more CUs busy (measured: tools/chainbench.py)
    if (d->weight_frag) {                                         // direct form: 32-row tiles, one wave per 32 couts
        a.xcd_tiles = (d->xcd_group_rows > 0 && !xcd_off && d->xcd_group_rows % 32 == 0 && d->rows % (8LL * d->xcd_group_rows) == 0) ? (int)(d->xcd_group_rows / 32) : 0;
        const bool tall = chain_direct_tall(d->C, d->rows);
        if (tall) a.xcd_tiles = a.xcd_tiles % 2 == 0 ? a.xcd_tiles / 2 : 0;
        if (d->nstage == 1) return launch_chain_direct<1>(a, d->C, tall, st);
        if (d->nstage == 2) return launch_chain_direct<2>(a, d->C, tall, st);
        return launch_chain_direct<3>(a, d->C, tall, st);
    }
    if (d->dtype == S2M2_F16) {
        switch (d->C) {
            case 128:
                return cfg == 's' ? launch_chain_n<half_t, 128, 32, 4>(a, d->nstage, st) : launch_chain_n<half_t, 128, 64, 4>(a, d->nstage, st);
            case 256:
                return cfg == 's' ? launch_chain_n<half_t, 256, 32, 8>(a, d->nstage, st) : launch_chain_n<half_t, 256, 64, 8>(a, d->nstage, st);
            case 384: return launch_chain_n<half_t, 384, 32, 4>(a, d->nstage, st);
            default: return launch_chain_n<half_t, 512, 32, 8>(a, d->nstage, st);
        }
    }
    if (d->C == 128) return launch_chain_n<float, 128, 32, 4>(a, d->nstage, st);
    return launch_chain_n<float, 256, 32, 8>(a, d->nstage, st);
}
extern "C" int s2m2_mlp_chain(const s2m2_chain_desc* d, void* stream) {
    return s2m2::plan_dispatch_desc<s2m2_chain_desc>("s2m2_mlp_chain", &mlp_chain_impl, d, stream);
}

