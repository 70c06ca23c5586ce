// K1 -- fused LayerNorm + all-pairs epipolar correlation (cost-volume construction).
//
// Replaces DispInit.forward's  layer_norm -> chunk -> einsum('...hic,...hjc->...hij')
// (/root/reference/src/s2m2/core/model/submodules.py:165,216-217).
//
// One image row (b,y) is an independent  w x w x C  contraction  S = LN(L) . LN(R)^T  (L, R: w tokens of C channels).
//   block  = NW waves = one image row (or a strip of NW*32 left pixels of it); NW is a launch parameter.
//   wave   = 32 left pixels AND 32 right pixels per chunk: at kernel entry it puts both token sets in flight with 16-B
//            coalesced loads (8 lanes per token) -- for w <= 32*NW that is the block's entire input, 2*w*C*sizeof(T)
//            bytes in flight per CU at once.  Tokens are LayerNorm'ed in fp32 (two-pass mean/var, 8-lane butterfly),
//            rounded to T and written to LDS rows padded by 16 B (conflict-free ds_read_b128 fragment reads).
//            The wave's own left tokens bounce through its LDS slice into MFMA fragments that stay in registers.
//   sync   = ONE block barrier per chunk of NW*32 right pixels (one per row when it fits); after it every wave
//            sweeps the column tiles on its own (staggered start), no further block-level synchronisation.
//   MFMA   = roles swapped on purpose: D = R_tile . L_wave^T, so a lane ends up with 4 CONSECUTIVE j of one row i
//            per register quad -> the accumulators go to a wave-private LDS tile with 4 wide writes per 32x32
//            tile (instead of 32 two-byte writes), come back as 16-B pieces of whole rows, and every global store
//            instruction writes 8 rows x 128 contiguous bytes of the cost volume.
// Every feature token is read from HBM once and normalised once per strip; the cost volume is written once.
// Algorithmic traffic per pair: 2*h*w*C*sizeof(T) read + h*w*w*sizeof(TO) written.
#include "common.h"
#include "plan.h"
#include <hip/hip_ext.h>
#include <stdlib.h>
#include <type_traits>

// compile-time ablation switches for tools/k1_ablate.py (never set in the shipped library):
// 1 no global stores, 2 no MFMA, 4 no LayerNorm arithmetic, 8 no multiply/store phase at all
#ifndef S2M2_LNCORR_DBG
#define S2M2_LNCORR_DBG 0
#endif

// timeline instrumentation for tools/k1_trace.py (experiment builds only: -DS2M2_LNCORR_TRACE=1): lane 0 of every wave stamps the
// shader clock at phase boundaries into a device array that s2m2_debug_k1_trace() copies out
#ifndef S2M2_LNCORR_TRACE
#define S2M2_LNCORR_TRACE 0
#endif
#if S2M2_LNCORR_TRACE
__device__ unsigned long long g_k1_trace[1024 * 16 * 16];
#define K1_T(slot)                                                                                                            \
    do {                                                                                                                      \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                                     \
            g_k1_trace[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memtime();         \
    } while (0)
#else
#define K1_T(slot)
#endif

namespace s2m2 {

template <typename T, typename TO, int C_, int NWCAP_, int RIF_, bool EARLY_B_, int TPW_ = 32, bool PRENORM_ = false>
struct LnCorrCfg {
    // PRENORM: the tokens arrive normalised (s2m2_corr: DispInit's LayerNorm was folded into the launch that produced them, the
    // second output of s2m2_mlp_chain) -- no statistics, no affine, the kernel is the batched R . L^T product and its stores
    static constexpr bool PRENORM = PRENORM_;
    static constexpr int C = C_;
    static constexpr int RIF = RIF_;                    // token rounds (8 tokens per wave each) kept in flight in registers
    static constexpr bool EARLY_B = EARLY_B_;           // right tokens of chunk 0 are requested together with the left ones
    // right tokens each wave normalises per chunk.  32: a chunk is as wide as the block's left strips (whole rows in one chunk at
    // C <= 128).  16 (wide C): the LDS a wave needs halves, so a block can hold a wave for EVERY left strip of a row and walk the
    // right row in chunks of 16*NW tokens -- the right row is read and normalised once per row instead of once per strip
    static constexpr int TPW = TPW_;
    static constexpr int ROUNDS = TPW / 8;              // 8-token rounds per wave per chunk
    static constexpr bool LEAN = TPW < 32 && sizeof(T) == 2;   // register-lean LayerNorm (normalize_store)
    static constexpr bool ALLRES = RIF == ROUNDS;       // every round of a chunk resident in registers: the next chunk is prefetched
    static constexpr int VEC = 16 / sizeof(T);          // elements per 16-B piece
    static constexpr int PIECES = C / VEC;              // pieces per token
    static constexpr int LPT = 8;                       // lanes per token
    static constexpr int PPL = PIECES / LPT;            // pieces per lane per token
    static constexpr int RS = C + VEC;                  // LDS row stride (elements): +16 B pad
    static constexpr int VECO = 16 / sizeof(TO);
    static constexpr int CRS = 64 + VECO;               // staging row stride (elements of TO): 2 tiles + 16 B pad
    static constexpr int KSTEPS = C / 16;
    static constexpr size_t GB_BYTES = 2 * C * sizeof(float);
    static constexpr size_t WB_BYTES = (size_t)TPW * RS * sizeof(T);     // per wave: its TPW normalised tokens
    static constexpr size_t WC_BYTES = (size_t)32 * CRS * sizeof(TO);    // per wave: 32 x 64 output staging
    static constexpr size_t lds_bytes(int nw) { return GB_BYTES + (size_t)nw * (WB_BYTES + WC_BYTES); }
    static constexpr int NWLDS = (int)((160 * 1024 - GB_BYTES) / (WB_BYTES + WC_BYTES));
    static constexpr int NWGRAN = 32 / TPW;             // waves per block come in multiples of this: chunks are whole 32-column tiles
    static constexpr int NWMAX = (NWLDS < NWCAP_ ? NWLDS : NWCAP_) / NWGRAN * NWGRAN;   // waves per block: LDS bound, register bound
    static_assert(PIECES % LPT == 0 && LPT == 8, "C must be a multiple of 8 pieces; group8_sum assumes 8 lanes per token");
    static_assert(NWMAX >= 1, "LDS budget");
    static_assert((TPW == 32 || TPW == 16) && ROUNDS % RIF == 0, "token rounds");
};

// LayerNorm (eps 1e-5, biased variance, affine) one token spread over 8 lanes (PPL 16-B pieces per lane) in fp32 and
// write it, rounded to T, to its LDS row.  Contraction is pinned (explicit fmaf, contract off) so that left and right
// tokens round identically: swapping the two images then transposes the cost volume bit-exactly (tests rely on it).
#pragma clang fp contract(off)
template <typename CFG, typename T>
__device__ __forceinline__ void normalize_store(const Vec16<T> (&p)[CFG::PPL], T* __restrict__ drow,
                                                const float* __restrict__ gb, int sub, int dbg = 0) {
    constexpr float inv_c = 1.0f / CFG::C;
    if (CFG::PRENORM || (dbg & 4)) {                     // normalised input (or ablation): plain copy into the padded LDS row
#pragma unroll
        for (int q = 0; q < CFG::PPL; ++q) *reinterpret_cast<Vec16<T>*>(drow + (sub + CFG::LPT * q) * CFG::VEC) = p[q];
        return;
    }
    if constexpr (CFG::LEAN) {
        // wide C: the fp32 copy of the token (C / 8 registers per lane) is what pushes these configurations over the register budget
        // of a block with one wave per left strip.  The three passes convert from the 16-bit pieces each time instead (the pieces are
        // made opaque in between, or the compiler would keep the converted values after all); same operations in the same order as
        // below, bit-identical results.
        Vec16<T> pp[CFG::PPL];
#pragma unroll
        for (int q = 0; q < CFG::PPL; ++q) pp[q] = p[q];
        auto opaque = [&]() {
#pragma unroll
            for (int q = 0; q < CFG::PPL; ++q) {
                raw16_t r = __builtin_bit_cast(raw16_t, pp[q]);
                asm volatile("" : "+v"(r));
                pp[q] = __builtin_bit_cast(Vec16<T>, r);
            }
        };
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < CFG::PPL; ++q)
#pragma unroll
            for (int e = 0; e < CFG::VEC; ++e) s += to_f32(pp[q].v[e]);
        const float mean = group8_sum(s) * inv_c;
        opaque();
        float ss = 0.f;
#pragma unroll
        for (int q = 0; q < CFG::PPL; ++q)
#pragma unroll
            for (int e = 0; e < CFG::VEC; ++e) { const float d = to_f32(pp[q].v[e]) - mean; ss = __builtin_fmaf(d, d, ss); }
        const float rstd = rsqrtf(group8_sum(ss) * inv_c + 1e-5f);
        opaque();
#pragma unroll
        for (int q = 0; q < CFG::PPL; ++q) {
            const int c0 = (sub + CFG::LPT * q) * CFG::VEC;
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < CFG::VEC; ++e)
                o.v[e] = from_f32<T>(__builtin_fmaf((to_f32(pp[q].v[e]) - mean) * rstd, gb[c0 + e], gb[CFG::C + c0 + e]));
            *reinterpret_cast<Vec16<T>*>(drow + c0) = o;
        }
        return;
    }
    float x[CFG::PPL][CFG::VEC];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < CFG::PPL; ++q)
#pragma unroll
        for (int e = 0; e < CFG::VEC; ++e) { x[q][e] = to_f32(p[q].v[e]); s += x[q][e]; }
    const float mean = group8_sum(s) * inv_c;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < CFG::PPL; ++q)
#pragma unroll
        for (int e = 0; e < CFG::VEC; ++e) { x[q][e] -= mean; ss = __builtin_fmaf(x[q][e], x[q][e], ss); }
    const float rstd = rsqrtf(group8_sum(ss) * inv_c + 1e-5f);
#pragma unroll
    for (int q = 0; q < CFG::PPL; ++q) {
        const int c0 = (sub + CFG::LPT * q) * CFG::VEC;
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < CFG::VEC; ++e) o.v[e] = from_f32<T>(__builtin_fmaf(x[q][e] * rstd, gb[c0 + e], gb[CFG::C + c0 + e]));
        *reinterpret_cast<Vec16<T>*>(drow + c0) = o;
    }
}

// fragment pick-up under a divergent branch: the read is tied to its destination registers ("+v"), so the lanes outside the
// branch keep their fragment without the compiler holding a second copy of the whole operand to select from.  The caller waits
// (s_waitcnt lgkmcnt(0)): the compiler does not track this read.
__device__ __forceinline__ void load_frag_keep(Frag<half_t>& f, const half_t* p) {
    raw16_t r = __builtin_bit_cast(raw16_t, f.v);
    asm volatile("ds_read_b128 %0, %1" : "+v"(r) : "v"(static_cast<unsigned>(reinterpret_cast<size_t>(p))));
    f.v = __builtin_bit_cast(half8_t, r);
}
__device__ __forceinline__ void load_frag_keep(Frag<float>& f, const float* p) { load_frag(f, p); }   // (fp32 configurations have one group)

template <typename CFG, typename T>
__device__ __forceinline__ void load_token(Vec16<T> (&p)[CFG::PPL], const T* __restrict__ src, int tok, int w, int sub) {
    tok = tok < w ? tok : w - 1;                                   // ragged tail: duplicates, never stored
    const T* q0 = src + (size_t)tok * CFG::C + sub * CFG::VEC;
#pragma unroll
    for (int q = 0; q < CFG::PPL; ++q) p[q] = *reinterpret_cast<const Vec16<T>*>(q0 + CFG::LPT * q * CFG::VEC);
}

template <typename TO> __device__ __forceinline__ void store_quad(TO* dst, float a, float b, float c, float d);
template <> __device__ __forceinline__ void store_quad<half_t>(half_t* dst, float a, float b, float c, float d) {
    half4_t v = {from_f32<half_t>(a), from_f32<half_t>(b), from_f32<half_t>(c), from_f32<half_t>(d)};
    *reinterpret_cast<half4_t*>(dst) = v;
}
template <> __device__ __forceinline__ void store_quad<float>(float* dst, float a, float b, float c, float d) {
    float4_t v = {a, b, c, d};
    *reinterpret_cast<float4_t*>(dst) = v;
}

// 16-byte store of a cost-volume piece.  The volume is written once and is larger than the L2 of the XCD that writes it; ``mode`` picks the
// cache policy of the store (S2M2_K1_NT, measured in profiles/r04/k1_store_modes.txt): 0 plain (write-back L2: up to 32 MB of dirty lines are
// flushed by the release at the end of the kernel, while no wave runs), 1 nt (streaming hint: slower), 2 sc1 = THE DEFAULT (write-through:
// the lines drain to the memory side while the kernel is still storing), 3 sc0 sc1 (same speed), 4 sc0 sc1 nt (slower)
template <typename TO>
__device__ __forceinline__ void store_cv(TO* dst, const Vec16<TO>& v, int mode) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 r = __builtin_bit_cast(f4, v);
    if (mode == 0) *reinterpret_cast<f4*>(dst) = r;
    else if (mode == 1) __builtin_nontemporal_store(r, reinterpret_cast<f4*>(dst));
    else if (mode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(r) : "memory");
    else if (mode == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(r) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(dst), "v"(r) : "memory");
}

template <typename CFG, typename T, typename TO>
__global__ __launch_bounds__(CFG::NWMAX * 64) void ln_corr_kernel(const T* __restrict__ feat, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, TO* __restrict__ cv,
                                                                 int B, int h, int w, int nstrip, int band, int pitch, int flags) {
    constexpr int dbg = S2M2_LNCORR_DBG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* gb = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = blockDim.x >> 6;
    T* Bs = reinterpret_cast<T*>(smem + CFG::GB_BYTES);                               // [NW*TPW][RS] normalised right tokens
    T* Wb = Bs + (size_t)wv * CFG::TPW * CFG::RS;                                     // this wave's TPW rows of it
    TO* Wc = reinterpret_cast<TO*>(smem + CFG::GB_BYTES + (size_t)NW * CFG::WB_BYTES + (size_t)wv * CFG::WC_BYTES);

    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int row = bid / nstrip;                       // (b, y)
    const int strip = bid - row * nstrip;
    const int b = row / h, y = row - b * h;
    const int i0 = (strip * NW + wv) * 32;              // first left pixel of this wave
    const bool wave_active = i0 < w;
    const T* left = feat + ((size_t)(b * h + y) * w) * CFG::C;
    const T* right = feat + ((size_t)((B + b) * h + y) * w) * CFG::C;
    TO* cvrow = cv + (size_t)row * w * pitch;           // volume rows `pitch` elements apart (>= w; 64-element multiples = whole 128-B lines)
    const int sub = lane & 7, trow = lane >> 3;
    const int TJ = NW * CFG::TPW;                       // right pixels per chunk (the whole row when w <= TJ); a multiple of 32
    const int nchunks = (w + TJ - 1) / TJ;
    K1_T(0);

    // ---- everything this wave needs first is put in flight at once: 32 left tokens (+ 32 right tokens of chunk 0)
    static_assert(!CFG::EARLY_B || CFG::ALLRES, "EARLY_B needs every round of a chunk in registers");
    // (the LayerNorm affine goes to LDS first: __syncthreads() drains vmcnt, so no token load may be pending across it)
    if constexpr (!CFG::PRENORM) {
        for (int c = tid; c < CFG::C; c += blockDim.x) { gb[c] = gamma[c]; gb[CFG::C + c] = beta[c]; }
        __syncthreads();
    }
    K1_T(1);
    Vec16<T> rawB[CFG::RIF][CFG::PPL];
    Frag<T> afrag[CFG::KSTEPS];
    // a group of TPW left tokens is complete in this wave's scratch rows: the lanes whose token (lane & 31) belongs to the group pick
    // their fragments up before the next group overwrites the rows
    auto pickup = [&](int grp) __attribute__((always_inline)) {
        __builtin_amdgcn_wave_barrier();
        if (grp == 4 / CFG::ROUNDS - 1) K1_T(3);
        const T* ap = Wb + (size_t)((lane & 31) % CFG::TPW) * CFG::RS + (lane >> 5) * 8;
        if (grp == 0) {                                           // every lane reads (the lanes of later groups pick up placeholders)
#pragma unroll
            for (int kk = 0; kk < CFG::KSTEPS; ++kk) load_frag(afrag[kk], ap + kk * 16);
        } else {
            if ((lane & 31) / CFG::TPW == grp) {
#pragma unroll
                for (int kk = 0; kk < CFG::KSTEPS; ++kk) load_frag_keep(afrag[kk], ap + kk * 16);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_wave_barrier();
    };
    // PRENORM: tokens go global -> registers -> LDS unchanged.  With nothing to compute on them the optimiser treats the staging
    // registers as a plain copy and sinks every load next to its LDS store -- one load in flight at a time (measured: 29 us against 22 us
    // WITH the LayerNorm).  So this path requests its tokens with untracked loads (common.h: global_load16_async) and counted waits.
    raw16_t pre_a[CFG::PRENORM ? CFG::RIF : 1][CFG::PPL], pre_b[CFG::PRENORM ? CFG::RIF : 1][CFG::PPL];
    auto pre_issue = [&](raw16_t (&dst)[CFG::PRENORM ? CFG::RIF : 1][CFG::PPL], const T* src, int tok0) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < CFG::RIF; ++r) {
            int tok = tok0 + r * 8 + trow;
            tok = tok < w ? tok : w - 1;                           // ragged tail: duplicates, never stored
            const T* q0 = src + (size_t)tok * CFG::C + sub * CFG::VEC;
#pragma unroll
            for (int q = 0; q < CFG::PPL; ++q) global_load16_async(dst[r][q], q0 + CFG::LPT * q * CFG::VEC);
        }
    };
    auto pre_stash = [&](raw16_t (&v)[CFG::PRENORM ? CFG::RIF : 1][CFG::PPL], int round0) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < CFG::RIF; ++r)
#pragma unroll
            for (int q = 0; q < CFG::PPL; ++q) {
                settle(v[r][q]);
                *reinterpret_cast<raw16_t*>(Wb + (size_t)(((round0 + r) * 8 + trow) % CFG::TPW) * CFG::RS + (sub + CFG::LPT * q) * CFG::VEC) = v[r][q];
            }
    };
    if constexpr (CFG::PRENORM) {
        pre_issue(pre_a, left, i0);
        if (CFG::EARLY_B) pre_issue(pre_b, right, wv * CFG::TPW);
        K1_T(2);
#pragma unroll
        for (int r0 = 0; r0 < 4; r0 += CFG::RIF) {
            if (r0 > 0) pre_issue(pre_a, left, i0 + r0 * 8);
            if (r0 == 0 && CFG::EARLY_B) wait_vmcnt<CFG::RIF * CFG::PPL>();      // the right tokens (requested later) may still be in flight
            else wait_vmcnt<0>();
            pre_stash(pre_a, r0);
            if ((r0 + CFG::RIF) % CFG::ROUNDS == 0) pickup((r0 + CFG::RIF) / CFG::ROUNDS - 1);
        }
        // the right tokens of chunk 0 (requested at entry) go to LDS HERE, not under a condition inside the chunk loop: an untracked
        // load whose destination is live across a control-flow merge may get a register copy there that reads it while in flight
        if (CFG::EARLY_B) {
            wait_vmcnt<0>();
            pre_stash(pre_b, 0);
        }
    } else {
        Vec16<T> rawA[CFG::RIF][CFG::PPL];
        // unconditional (token indices are clamped): a branch here lets the compiler hoist the first LayerNorm under it and push
        // the right-token requests behind the arrival of the left tokens
#pragma unroll
        for (int r = 0; r < CFG::RIF; ++r) load_token<CFG, T>(rawA[r], left, i0 + r * 8 + trow, w, sub);
        if (CFG::EARLY_B) {
#pragma unroll
            for (int r = 0; r < CFG::RIF; ++r)
                load_token<CFG, T>(rawB[r], right, wv * CFG::TPW + r * 8 + trow, w, sub);
        }
        // keep every request above in front of the arithmetic below: without this the scheduler sinks the right-token loads under
        // the LayerNorm of the left tokens (one full memory latency lost)
        __builtin_amdgcn_sched_barrier(0);
        K1_T(2);
        // left operand: normalise -> (this wave's slice of Bs as scratch) -> k16 fragments in registers.  Unconditional (clamped
        // duplicates for a wave past the row end): any branch between the requests and their first use lets the optimiser sink
        // the loads into it, behind the other requests and behind the scheduling barrier above.
        // TPW-row scratch: the 32 tokens pass through it in 32 / TPW groups
#pragma unroll
        for (int r0 = 0; r0 < 4; r0 += CFG::RIF) {
            if (r0 > 0) {
#pragma unroll
                for (int r = 0; r < CFG::RIF; ++r) load_token<CFG, T>(rawA[r], left, i0 + (r0 + r) * 8 + trow, w, sub);
            }
#pragma unroll
            for (int r = 0; r < CFG::RIF; ++r)
                normalize_store<CFG, T>(rawA[r], Wb + (size_t)(((r0 + r) * 8 + trow) % CFG::TPW) * CFG::RS, gb, sub, dbg);
            if ((r0 + CFG::RIF) % CFG::ROUNDS == 0) pickup((r0 + CFG::RIF) / CFG::ROUNDS - 1);
        }
    }

    for (int c = 0; c < nchunks; ++c) {
        // right tokens of this chunk: each wave normalises its 32 into its slice, then prefetches its share of the next chunk
        const int t0 = c * TJ + wv * CFG::TPW;
        if constexpr (CFG::PRENORM) {
            // (no prefetch across the multiply / store phase here: a counted wait would also wait for this wave's stores -- they tick
            // the same counter -- so a chunk's tokens are requested at its top; rows that fit one chunk, C <= 128, are unaffected)
            if (!CFG::EARLY_B || c > 0) {                         // (chunk 0 of the EARLY_B configurations is in LDS already)
#pragma unroll
                for (int r0 = 0; r0 < CFG::ROUNDS; r0 += CFG::RIF) {
                    pre_issue(pre_b, right, t0 + r0 * 8);
                    wait_vmcnt<0>();
                    pre_stash(pre_b, r0);
                }
            }
        } else {
#pragma unroll
        for (int r0 = 0; r0 < CFG::ROUNDS; r0 += CFG::RIF) {
            if (!(CFG::ALLRES && (c > 0 || CFG::EARLY_B))) {
#pragma unroll
                for (int r = 0; r < CFG::RIF; ++r) load_token<CFG, T>(rawB[r], right, t0 + (r0 + r) * 8 + trow, w, sub);
            }
#pragma unroll
            for (int r = 0; r < CFG::RIF; ++r)
                normalize_store<CFG, T>(rawB[r], Wb + (size_t)((r0 + r) * 8 + trow) * CFG::RS, gb, sub, dbg);
        }
        if (CFG::ALLRES && c + 1 < nchunks) {
#pragma unroll
            for (int r = 0; r < CFG::RIF; ++r) load_token<CFG, T>(rawB[r], right, t0 + TJ + r * 8 + trow, w, sub);
        }
        }
        K1_T(4);
        __syncthreads();
        K1_T(5);
        if (wave_active && !(dbg & 8)) {
            const int jbase = c * TJ;
            int ntile = (w - jbase + 31) / 32;                    // 32-wide column tiles with data in this chunk
            ntile = ntile < TJ / 32 ? ntile : TJ / 32;
            int npair = (ntile + 1) >> 1;
            // banded volume (band >= 0, use_positivity models): only columns j <= i + band are ever read downstream (the masked
            // Sinkhorn, the +-4 tap lookups at disparities >= 0), so this wave (rows i0 .. i0+31) stops at column i0 + 31 + band
            if (band >= 0) {
                const int last_col = i0 + 31 + band - jbase;
                const int need = last_col < 0 ? 0 : last_col / 64 + 1;
                npair = npair < need ? npair : need;
                const int nt = 2 * npair;
                ntile = ntile < nt ? ntile : nt;
            }
            if (npair > 0) {
            // The column tiles are taken two at a time (64 columns = 128-B row segments in the stores), starting at a
            // different pair per wave.  Software pipeline: the 16 MFMAs of pair p+1 (two independent accumulators,
            // interleaved) are issued BEFORE pair p is read back from the staging tile and stored, so the matrix
            // pipe works while this wave sits in the (HBM-bound) store queue.
            float16_t acc0, acc1;
            auto pair_of = [&](int pp) { int pr = pp + wv; pr = pr >= npair ? pr - npair : pr; return pr >= npair ? pr % npair : pr; };
            auto mma_pair = [&](int pr) {
                const int ct0 = pr * 2;
                const bool two = ct0 + 1 < ntile;
                const T* bp0 = Bs + (size_t)(ct0 * 32 + (lane & 31)) * CFG::RS + (lane >> 5) * 8;
                const T* bp1 = two ? bp0 + 32 * CFG::RS : bp0;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
                if (!(dbg & 2)) {
#pragma unroll
                    for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
                        Frag<T> b0, b1;
                        load_frag(b0, bp0 + kk * 16);
                        load_frag(b1, bp1 + kk * 16);
                        mma32(acc0, b0, afrag[kk]);               // D[j][i]: lane = left pixel i, registers = right pixels j
                        mma32(acc1, b1, afrag[kk]);
                    }
                }
            };
            mma_pair(pair_of(0));
            for (int pp = 0; pp < npair; ++pp) {
                const int pr = pair_of(pp);
                TO* wrow = Wc + (size_t)(lane & 31) * CFG::CRS + 4 * (lane >> 5);
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    store_quad<TO>(wrow + 8 * g, acc0[4 * g], acc0[4 * g + 1], acc0[4 * g + 2], acc0[4 * g + 3]);
                    store_quad<TO>(wrow + 32 + 8 * g, acc1[4 * g], acc1[4 * g + 1], acc1[4 * g + 2], acc1[4 * g + 3]);
                }
                __builtin_amdgcn_wave_barrier();
                if (pp + 1 < npair) mma_pair(pair_of(pp + 1));
                constexpr int PPR = 64 / CFG::VECO;               // 16-B pieces per staged row (64 columns)
                constexpr int ITERS = 32 * PPR / 64;
                const int j0 = jbase + pr * 64;
#pragma unroll
                for (int it = 0; it < ITERS; ++it) {
                    const int q = it * 64 + lane;
                    const int rr = q / PPR, pc = q - rr * PPR;
                    const int i = i0 + rr;
                    const int j = j0 + pc * CFG::VECO;
                    const Vec16<TO> v = *reinterpret_cast<const Vec16<TO>*>(Wc + rr * CFG::CRS + pc * CFG::VECO);
                    if (i < w && j < w && !(dbg & 1)) store_cv(cvrow + (size_t)i * pitch + j, v, flags);
                }
                __builtin_amdgcn_wave_barrier();
                K1_T(6 + (pp < 8 ? pp : 8));
            }
            }
        }
        K1_T(15);
        if (c + 1 < nchunks) __syncthreads();                     // all readers done before Bs is overwritten
    }
}

// ------------------------------------------------------------------------------------------------
// host dispatch
// ------------------------------------------------------------------------------------------------
// options of one launch (s2m2_corr_desc): nothing is passed through globals -- every entry point fills this struct
struct K1Opt {
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;   // recorded on the dispatch itself (hipExtLaunchKernel): at the start / end of the kernel's execution
    int band = -1;                                      // columns right of the diagonal that must be valid (-1: all)
    int pitch = 0;                                      // elements between volume rows (0: w)
};

template <typename CFG, typename T, typename TO>
static int launch_ln_corr(const void* feat, const float* g, const float* bta, void* cv, int B, int h, int w, const K1Opt& o, hipStream_t st);

// normalised-input variant of a configuration (same tiling)
template <typename CFG> struct PrenormOf;
template <typename T, typename TO, int C, int NWCAP, int RIF, bool EB, int TPW>
struct PrenormOf<LnCorrCfg<T, TO, C, NWCAP, RIF, EB, TPW, false>> { using type = LnCorrCfg<T, TO, C, NWCAP, RIF, EB, TPW, true>; };

template <typename CFG, typename T, typename TO>
static int launch_ln_corr(const void* feat, const float* g, const float* bta, void* cv, int B, int h, int w, const K1Opt& o, hipStream_t st) {
    auto kern = ln_corr_kernel<CFG, T, TO>;
    static size_t lds_granted[kMaxDevices] = {};                     // per instantiation
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::lds_bytes(CFG::NWMAX), lds_granted, "ln_corr")) return 1;
    const int tiles = (w + 31) / 32;                     // 32-pixel row tiles = waves needed per image row
    int nstrip = (tiles + CFG::NWMAX - 1) / CFG::NWMAX;
    // small problems: split rows into strips until the grid covers the chip (each strip re-reads / re-normalises the right row).  More, smaller
    // blocks do not help the 120-row case (c2): 240 three-wave blocks 12.1 us, 480-600 one- / two-wave blocks 14.6-15.6 us (profiles/r05/k1_minblocks.txt)
    while (B * h * nstrip < 200 && nstrip < tiles && (tiles + nstrip) / (nstrip + 1) >= 2) ++nstrip;
    int nw = (tiles + nstrip - 1) / nstrip;
    nw = (nw + CFG::NWGRAN - 1) / CFG::NWGRAN * CFG::NWGRAN;      // chunks of nw * TPW right tokens are whole 32-column tiles
    const int nblocks = B * h * nstrip;
    const int pitch = o.pitch > 0 ? o.pitch : w;
    // cache policy of the volume stores (store_cv): sc1 write-through by default -- measured (profiles/r04/k1_store_modes.txt, ab_k1_store_sc1lib_overlap.txt)
    // 19.2 -> 16.8 us back to back and 19.7-20.7 -> 17.7 us inside the forward against the write-back default of rounds 1-3; S2M2_K1_NT=0..4 A/B
    // (only for volume rows on 128-byte lines: a write-through store of a PARTIAL line is a read-modify-write at the memory side -- dense 608-byte
    // rows measured 25.0 us with sc1 against 21.8 with plain stores, profiles/r04/kbench.txt; the engine always allocates aligned rows)
    static const int k1_env = getenv("S2M2_K1_NT") ? atoi(getenv("S2M2_K1_NT")) : -1;
    const int k1_flags = k1_env >= 0 ? k1_env : ((pitch * (int)sizeof(TO)) % 128 == 0 ? 2 : 0);
    if (o.ev_start || o.ev_stop)
        hipExtLaunchKernelGGL(kern, dim3(nblocks), dim3(nw * 64), CFG::lds_bytes(nw), st, o.ev_start, o.ev_stop, 0,
                              static_cast<const T*>(feat), g, bta, static_cast<TO*>(cv), B, h, w, nstrip, o.band, pitch, k1_flags);
    else
        hipLaunchKernelGGL(kern, dim3(nblocks), dim3(nw * 64), CFG::lds_bytes(nw), st,
                           static_cast<const T*>(feat), g, bta, static_cast<TO*>(cv), B, h, w, nstrip, o.band, pitch, k1_flags);
    return check_launch("ln_corr");
}

// per (dtype, C): cap on waves per block from the registers the resident left-operand fragments need (the LDS bound --
// 32 normalised tokens + a 32x64 staging tile per wave -- is computed in LnCorrCfg)
template <typename T, int C> struct LnCorrPick;
template <> struct LnCorrPick<half_t, 64>  { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 64, 12, 4, true>; };
template <> struct LnCorrPick<half_t, 128> { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 128, 11, 4, true>; };
template <> struct LnCorrPick<half_t, 192> { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 192, 10, 2, true, 16>; };
template <> struct LnCorrPick<half_t, 256> { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 256, 10, 2, true, 16>; };
template <> struct LnCorrPick<half_t, 384> { template <typename TO> using cfg = LnCorrCfg<half_t, TO, 384, 8, 2, true, 16>; };
template <> struct LnCorrPick<float, 64>   { template <typename TO> using cfg = LnCorrCfg<float, TO, 64, 12, 4, true>; };
template <> struct LnCorrPick<float, 128>  { template <typename TO> using cfg = LnCorrCfg<float, TO, 128, 8, 4, false>; };
template <> struct LnCorrPick<float, 192>  { template <typename TO> using cfg = LnCorrCfg<float, TO, 192, 8, 2, false>; };
template <> struct LnCorrPick<float, 256>  { template <typename TO> using cfg = LnCorrCfg<float, TO, 256, 4, 2, false>; };
template <> struct LnCorrPick<float, 384>  { template <typename TO> using cfg = LnCorrCfg<float, TO, 384, 4, 1, false>; };

template <typename T, typename TO, bool PRENORM = false>
static int dispatch_c(const void* feat, const float* g, const float* bta, void* cv, int B, int h, int w, int C, const K1Opt& o, hipStream_t st) {
    switch (C) {
#define S2M2_CASE(CC)                                                                                                        \
    case CC: {                                                                                                               \
        using BASE = typename LnCorrPick<T, CC>::template cfg<TO>;                                                           \
        using CFG = typename std::conditional<PRENORM, typename PrenormOf<BASE>::type, BASE>::type;                          \
        return launch_ln_corr<CFG, T, TO>(feat, g, bta, cv, B, h, w, o, st);                                                 \
    }
        S2M2_CASE(64) S2M2_CASE(128) S2M2_CASE(192) S2M2_CASE(256) S2M2_CASE(384)
#undef S2M2_CASE
        default: return set_error("ln_corr: unsupported channel count C=%d (supported: 64,128,192,256,384)", C);
    }
}

}  // namespace s2m2

static int cost_volume_impl(const s2m2_corr_desc* d, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(d, "cost_volume: null descriptor");
    S2M2_REQUIRE(d->tokens && d->cv, "cost_volume: null pointer");
    S2M2_REQUIRE((d->ln_weight == nullptr) == (d->ln_bias == nullptr), "cost_volume: ln_weight and ln_bias come together (both null: the tokens are normalised already)");
    S2M2_REQUIRE(d->B > 0 && d->h > 0 && d->w > 0, "cost_volume: bad shape B=%d h=%d w=%d", d->B, d->h, d->w);
    S2M2_REQUIRE(d->w % 8 == 0, "cost_volume: w=%d must be a multiple of 8 (image width multiple of 32)", d->w);
    const int pitch = d->cv_pitch == 0 ? d->w : d->cv_pitch;
    S2M2_REQUIRE(pitch >= d->w && pitch % 8 == 0, "cost_volume: cv_pitch=%d must be a multiple of 8 and at least w=%d", d->cv_pitch, d->w);
    S2M2_REQUIRE(d->band >= -1, "cost_volume: band=%d (-1: the full volume, >= 0: columns j <= i + band)", d->band);
    K1Opt o;
    o.ev_start = static_cast<hipEvent_t>(d->start_event);
    o.ev_stop = static_cast<hipEvent_t>(d->stop_event);
    o.band = d->band;
    o.pitch = pitch;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int td = d->token_dtype, cd = d->cv_dtype;
    if (d->ln_weight) {
        if (td == S2M2_F16 && cd == S2M2_F16) return dispatch_c<half_t, half_t>(d->tokens, d->ln_weight, d->ln_bias, d->cv, d->B, d->h, d->w, d->C, o, st);
        if (td == S2M2_F16 && cd == S2M2_F32) return dispatch_c<half_t, float>(d->tokens, d->ln_weight, d->ln_bias, d->cv, d->B, d->h, d->w, d->C, o, st);
        if (td == S2M2_F32 && cd == S2M2_F32) return dispatch_c<float, float>(d->tokens, d->ln_weight, d->ln_bias, d->cv, d->B, d->h, d->w, d->C, o, st);
    } else {
        if (td == S2M2_F16 && cd == S2M2_F16) return dispatch_c<half_t, half_t, true>(d->tokens, nullptr, nullptr, d->cv, d->B, d->h, d->w, d->C, o, st);
        if (td == S2M2_F16 && cd == S2M2_F32) return dispatch_c<half_t, float, true>(d->tokens, nullptr, nullptr, d->cv, d->B, d->h, d->w, d->C, o, st);
        if (td == S2M2_F32 && cd == S2M2_F32) return dispatch_c<float, float, true>(d->tokens, nullptr, nullptr, d->cv, d->B, d->h, d->w, d->C, o, st);
    }
    return set_error("cost_volume: unsupported dtype pair tokens=%d cv=%d", td, cd);
}
extern "C" int s2m2_cost_volume(const s2m2_corr_desc* d, void* stream) {
    return s2m2::plan_dispatch_desc<s2m2_corr_desc>("s2m2_cost_volume", &cost_volume_impl, d, stream);
}


extern "C" int s2m2_event_create(void** event) {
    S2M2_REQUIRE(event, "event_create: null pointer");
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return s2m2::set_error("event_create: hipEventCreate failed");
    *event = e;
    return 0;
}

extern "C" int s2m2_event_destroy(void* event) {
    if (event && hipEventDestroy(static_cast<hipEvent_t>(event)) != hipSuccess) return s2m2::set_error("event_destroy: hipEventDestroy failed");
    return 0;
}

extern "C" int s2m2_event_elapsed_us(void* start_event, void* stop_event, float* microseconds) {
    S2M2_REQUIRE(start_event && stop_event && microseconds, "event_elapsed_us: null pointer");
    float ms = 0.f;
    const hipError_t e = hipEventElapsedTime(&ms, static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event));
    if (e != hipSuccess) return s2m2::set_error("event_elapsed_us: %s", hipGetErrorString(e));
    *microseconds = ms * 1e3f;
    return 0;
}

#if S2M2_LNCORR_TRACE
extern "C" int s2m2_debug_k1_trace(void* host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_k1_trace), bytes < sizeof(g_k1_trace) ? bytes : sizeof(g_k1_trace)) == hipSuccess ? 0 : 1;
}
#endif

extern "C" const char* s2m2_ln_corr_kernel_name(int C, int feat_dtype, int cv_dtype) {
    (void)C; (void)feat_dtype; (void)cv_dtype;
    return "ln_corr_kernel";
}
