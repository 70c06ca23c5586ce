// K13 -- one whole 1-D (epipolar) attention step of the multi-resolution transformer in ONE launch (s2m2_row_attn).
//
// Reference: BasicAttnBlock.forward (attentions.py:347-355) applies, to the token rows of the left and right feature maps,
//     z = z + proj(attn(LN(z)));   z = z + ffn(LN(z))        twice: CrossAttnBlock1D (:131-161, keys / values from the OTHER view's row,
// shared weights, both directions) + FFN (:229-250), then SelfAttnBlock1D (:99-128) + FFN.  Every operation of such a step is local to one
// image row (the epipolar line) and its partner row in the other view.  As separate launches (K9 fan-out for Q | K | V, K4, K9 chain) a step
// moves Q, K, V and the attention output through HBM (160 MB per direction pair at 256 x 304 x 128) and pays three launch boundaries; here
//
//   block = ONE token row (image b, line y): w tokens x 128 channels, one wave per 32 tokens, every wave keeps ITS tokens in registers from
//           the pre-LayerNorm to the store of the block's result -- Q, the attention output, proj, both FFN layers never leave the wave;
//   K / V  of the source row (the partner row for cross attention, the row itself for self attention) are projected by the block itself,
//           160 keys at a time, into LDS (K as MFMA A-fragments row-major, V transposed), and consumed by all waves (online softmax, the
//           same per-32-key update as K4, log2 domain);
//   weights: each 128 x 128 layer is copied once per block into LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write
//           pass; three 32 KB buffers, the K buffer doubles as the third) -- the next phase's layers land while the current phase computes --
//           and read as A-fragments with one ds_read_b128 per MFMA.  The layers arrive UNIT-MAJOR (16-byte unit u of row r at unit index
//           u * 128 + r: include/s2m2_hip.h "row_attn packing"), so the DMA is a linear 32 KB copy (1 KB per wave instruction, whole cache
//           lines), the 16 lanes of a fragment read hit 16 consecutive units (no bank conflict, no padding, no swizzle) and every fragment
//           address is the lane's base plus an immediate (an XOR swizzle of row-major rows cost 24 address registers: 70 spilled).
//
// Register-resident activations: the accumulator layout of v_mfma_f32_32x32x16 (lane = token, 16 channels {8j + 4hi + i}) rounded to fp16 IS a
// valid B operand of the next layer if the contraction index is relabelled consistently on both operands: k-slot (16s + 8hi + e) of step s holds
// channel 16s + 4hi + e (e < 4) / 16s + 8 + 4hi + (e - 4).  The weights arrive with the matching permutation of their columns (two 8-byte
// quads swapped per 16 columns: the "row_attn packing" of include/s2m2_hip.h, applied once per layer by the caller), K is written by the lanes that computed it in the same order, so no activation is ever transposed or
// exchanged between lanes.  V is computed with swapped MFMA operands (lane = channel, registers = tokens), which is its transposed layout.
// Rounding points are those of the separate launches: Q, K, V, probabilities, attention output, proj output, the residual sum, the FFN hidden
// tensor, the FFN output and the final sum are each rounded to fp16 once; softmax statistics, LayerNorm statistics and accumulators are fp32.
//
// fp16, C = 128 (the S model's 1/4 and 1/8 levels), heads 1 or 2, rows of up to 320 tokens.
#include "common.h"
#include "plan.h"
#include "epilogue.h"
#include <math.h>
#include <stdlib.h>

namespace s2m2 {

namespace ra {
constexpr int C = 128;
constexpr int WRS = C + 8;                      // K row stride in LDS (elements): 272 bytes, conflict-free b128 reads
constexpr int KC = 160;                         // keys per chunk (5 tiles of 32)
constexpr int VRS = KC + 8;                     // Vt row stride (elements): 336 bytes, conflict-free b128 reads
constexpr int W_BYTES = C * C * 2;              // 32768: a layer unit-major (16 column units x 128 rows x 16 bytes)
constexpr int K_BYTES = KC * WRS * 2;           // 43520 (>= W_BYTES: the K buffer doubles as the third weight buffer)
constexpr int V_BYTES = C * VRS * 2;            // 43008
constexpr int OFF_W0 = 0, OFF_W1 = W_BYTES, OFF_K = 2 * W_BYTES, OFF_V = OFF_K + K_BYTES, OFF_ST = OFF_V + V_BYTES;
constexpr int MAXW = 10;                        // waves per block = 32-token tiles per row
constexpr int OFF_CV = OFF_ST + MAXW * 32 * 8;  // per-channel vectors of the running phase: 6 x 128 floats
constexpr int LDS_BYTES = OFF_CV + 6 * C * 4;
static_assert(LDS_BYTES <= 160 * 1024, "row attention: LDS budget");
}  // namespace ra

// timeline instrumentation for tools/rowattn_trace.py (experiment builds only: -DS2M2_RA_TRACE=1): shader-clock stamps per phase and wave
#ifndef S2M2_RA_TRACE
#define S2M2_RA_TRACE 0
#endif
#if S2M2_RA_TRACE
__device__ unsigned long long g_ra_trace[1024 * 10 * 16];
#define RA_T(slot)                                                                                                     \
    do {                                                                                                               \
        if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024)                                                              \
            g_ra_trace[((size_t)blockIdx.x * 10 + (threadIdx.x >> 6)) * 16 + (slot)] = __builtin_amdgcn_s_memtime();  \
    } while (0)
#else
#define RA_T(slot) do {} while (0)
#endif

struct RowAttnArgs {
    const half_t* x; half_t* out; half_t* ln_out;
    long long xs, os, ls;                       // elements between tokens
    int nimg, h, w, heads, src_off, xcd_rows;
    const half_t* wgt;                          // q, k, v, proj, ffn.0, ffn.2: six 128 x 128 layers in the row_attn packing, back to back
    const float* vec;                           // twelve fp32 vectors of 128 (s2m2_rowattn_desc.vectors)
    float ln_eps, ln_out_eps, scale;
    const void* zero;
    int dbg;                                    // S2M2_RA_DBG (timing ablations, wrong results): 1 no attention loop, 2 no softmax arithmetic,
};                                              // 4 no K / V projection, 8 no tail layers, 16 no MFMAs in the attention loop

typedef float raw8_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ raw8_t global_load8(const void* p) { return *(const __attribute__((address_space(1))) raw8_t*)(p); }

// Per-channel fp32 vectors (bias, row sums, LayerNorm affine) in the lane = token accumulator layout: quad g of tile T = channels
// 32T + 8g + 4hi .. +3 -- two addresses per wave.  The vectors of a phase are staged in LDS (6 x 512 bytes) and read as broadcast 16-byte
// pieces right where they are used: as vector loads from global memory they were either hoisted in front of the MFMAs (128 registers) or, with
// the half selected by address, 546 single-dword loads per wave; as scalar loads they spilled 700 SGPRs.
__device__ __forceinline__ void ra_cvec(float (&v)[4], const float* lds_vec, int co, int hi) {
    const float4_t u = *reinterpret_cast<const float4_t*>(lds_vec + co + 4 * hi);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = u[e];
}
// Block barrier that orders LDS traffic only.  __syncthreads() in a kernel with LDS-DMA in flight also drains vmcnt (hipcc cannot tell the DMA
// from other vector-memory operations): every scratch store and ordinary prefetch in flight is waited for at the barrier (6 us at the
// first chunk's barrier: profiles/r06/rowattn_trace_first.txt).  Used where no DMA'd data is consumed behind the barrier.
__device__ __forceinline__ void ra_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// N consecutive per-channel fp32 vectors (global) -> LDS rows of 128 floats by LDS-DMA: wave 0, 1 KB (two vectors) per instruction
template <int N>
__device__ __forceinline__ void ra_dma_vecs(float* dst, const float* src, int wv, int lane) {
    static_assert(N % 2 == 0, "pairs of vectors");
    if (wv != 0) return;
#pragma unroll
    for (int j = 0; j < N / 2; ++j) {
        int ln = lane;
        asm volatile("" : "+v"(ln));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 2 * ra::C + ln * 4),
                                         (__attribute__((address_space(3))) void*)(dst + j * 2 * ra::C), 16, 0, 0);
    }
}

// 32 tokens x 128 channels of one wave as 8 B-operand fragments in the relabelled k order (see the header): fragment s of lane (token, hi) =
// channels 16s + 4hi .. +3 and 16s + 8 + 4hi .. +3
typedef Frag<half_t> Tile[8];

__device__ __forceinline__ void ra_load_tile(Tile& a, const half_t* tok, int hi) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const raw8_t lo = global_load8(tok + 16 * s + 4 * hi), up = global_load8(tok + 16 * s + 8 + 4 * hi);
        const raw16_t v = {lo.x, lo.y, up.x, up.y};
        a[s].v = __builtin_bit_cast(half8_t, v);
    }
}
// the same tile requested with loads the compiler does not track (no s_waitcnt of its own: with LDS-DMA in flight hipcc would wait for
// vmcnt(0), i.e. for every DMA issued after them as well); consumed through ra_settle_tile after a counted wait
struct RawTile { raw8_t r[16]; };
__device__ __forceinline__ void ra_load_tile_async(RawTile& t, const half_t* tok, int hi) {
    const half_t* p = tok + 4 * hi;
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(t.r[k]) : "v"(p), "n"(k * 16));
}
__device__ __forceinline__ void ra_settle_tile(Tile& a, RawTile& t) {
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" : "+v"(t.r[k]));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const raw16_t v = {t.r[2 * s].x, t.r[2 * s].y, t.r[2 * s + 1].x, t.r[2 * s + 1].y};
        a[s].v = __builtin_bit_cast(half8_t, v);
    }
}
__device__ __forceinline__ void ra_wait_vmcnt(int n) {           // n outstanding vector-memory operations may remain (block-uniform n <= 16)
    switch (n) {
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

__device__ __forceinline__ void ra_store_tile(half_t* tok, const Tile& a, int hi) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const raw16_t v = __builtin_bit_cast(raw16_t, a[s].v);
        *reinterpret_cast<raw8_t*>(tok + 16 * s + 4 * hi) = raw8_t{v.x, v.y};
        *reinterpret_cast<raw8_t*>(tok + 16 * s + 8 + 4 * hi) = raw8_t{v.z, v.w};
    }
}

// LayerNorm statistics (no affine) of the lane's token: each half-wave holds 64 of the 128 channels
__device__ __forceinline__ void ra_ln_stats(const Tile& a, float eps, float& mean, float& rstd) {
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) ln_accumulate(a[k], s, q, 0.f);
    s += __shfl_xor(s, 32);
    q += __shfl_xor(q, 32);
    const float inv = 1.0f / (float)ra::C;
    mean = s * inv;
    rstd = rsqrtf(fmaxf(__builtin_fmaf(-mean, mean, q * inv), 0.f) + eps);
}

// One 128 x 128 fp16 layer (row_attn packing, global) -> an LDS weight buffer by LDS-DMA: a linear copy, 1 KB per wave instruction.
// Asynchronous: the data is in LDS after the issuing wave's next vmcnt(0), visible to the block after the barrier behind it.
__device__ __forceinline__ void ra_dma(const half_t* W, half_t* dst, int wv, int nwv, int lane) {
#pragma unroll 1
    for (int i = wv; i < 32; i += nwv) {
        int ln = lane;
        asm volatile("" : "+v"(ln));                             // the per-lane address is formed HERE: hoisted, one 64-bit address per layer
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(W + i * 512 + ln * 8),   // (12 registers) lives in scratch
                                         (__attribute__((address_space(3))) void*)(dst + i * 512), 16, 0, 0);
    }
}

// acc[T] = W[32T .. 32T+31][:] . a   (SWAP: a . W^T -- lane = output channel, registers = tokens)
template <bool SWAP>
__device__ __forceinline__ void ra_linear(float16_t (&acc)[4], const half_t* Wl, const Tile& a, int l31, int hi) {
#pragma unroll
    for (int T = 0; T < 4; ++T)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[T][r] = 0.f;
    const half_t* wp = Wl + (hi * ra::C + l31) * 8;              // unit (2s + hi) of row 32T + l31: unit index (2s + hi) * 128 + row
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int T = 0; T < 4; ++T) {
            Frag<half_t> wf;
            load_frag(wf, wp + s * (2 * ra::C * 8) + T * 32 * 8);
            if constexpr (SWAP) mma32(acc[T], a[s], wf);
            else mma32(acc[T], wf, a[s]);
        }
    }
}

// one 32-channel output tile at a time (the projection phase runs next to the attention state: 96 live registers, 168 per wave at 10 waves)
template <bool SWAP>
__device__ __forceinline__ void ra_linear_tile(float16_t& acc, const half_t* Wl, const Tile& a, int T, int l31, int hi) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const half_t* wp = Wl + (hi * ra::C + T * 32 + l31) * 8;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        Frag<half_t> wf;
        load_frag(wf, wp + s * (2 * ra::C * 8));
        if constexpr (SWAP) mma32(acc, a[s], wf);
        else mma32(acc, wf, a[s]);
    }
}

// bias / folded LayerNorm / activation of a lane = token accumulator set -> fp16 tile in the same registers' layout (stage_tile's arithmetic)
template <int ACT, bool LN>
__device__ __forceinline__ void ra_epilogue(Tile& out, const float16_t (&acc)[4], const float* bias, const float* wsum,
                                            float mean, float rstd, int hi) {
#pragma unroll
    for (int T = 0; T < 4; ++T) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int co = 32 * T + 8 * g;
            float bv[4], v[4];
            ra_cvec(bv, bias, co, hi);
            if constexpr (LN) {
                float ws[4];
                ra_cvec(ws, wsum, co, hi);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(rstd, __builtin_fmaf(-mean, ws[e], acc[T][4 * g + e]), bv[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[T][4 * g + e] + bv[e];
            }
            if constexpr (ACT == S2M2_ACT_GELU) {
                const float2_t r0 = fast_gelu16x2((float2_t){v[0], v[1]}), r1 = fast_gelu16x2((float2_t){v[2], v[3]});
                v[0] = r0.x; v[1] = r0.y; v[2] = r1.x; v[3] = r1.y;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) out[2 * T + (g >> 1)].v[4 * (g & 1) + e] = from_f32<half_t>(v[e]);
        }
    }
}

__device__ __forceinline__ void ra_add(Tile& a, const Tile& b) {        // a = fp16(a + b), element-wise in fp32 (the residual sums)
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[s].v[e] = from_f32<half_t>((float)a[s].v[e] + (float)b[s].v[e]);
}

template <int MAXT, int H>
__global__ __launch_bounds__(MAXT) void row_attn_kernel(RowAttnArgs a) {
    using namespace ra;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* W0 = reinterpret_cast<half_t*>(smem + OFF_W0);
    half_t* W1 = reinterpret_cast<half_t*>(smem + OFF_W1);
    half_t* Ks = reinterpret_cast<half_t*>(smem + OFF_K);
    half_t* Vt = reinterpret_cast<half_t*>(smem + OFF_V);
    const int tid = threadIdx.x, nthr = blockDim.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwv = nthr >> 6;
    float2_t* stat = reinterpret_cast<float2_t*>(smem + OFF_ST) + wv * 32;
    float* cv = reinterpret_cast<float*>(smem + OFF_CV);         // front: vectors 0..5 (q, k, v: bias, row sums);  tail: vectors 6..11

    // row of this block; with the placement hint, XCD x (blocks are dealt round robin) owns the x-th eighth of every image's lines, and the
    // rows of one line in all images (a cross-attention pair included) are neighbours in dispatch order on that XCD
    int row = blockIdx.x;
    if (a.xcd_rows > 0) {
        const int x = row & 7, k = row >> 3;
        const int yl = k / a.nimg, img = k - yl * a.nimg;
        row = img * a.h + x * a.xcd_rows + yl;
    }
    const int rows = a.nimg * a.h;
    int srow = row + a.src_off;
    srow = srow >= rows ? srow - rows : srow;
    const half_t* xrow = a.x + (long long)row * a.w * a.xs;
    const half_t* srcrow = a.x + (long long)srow * a.w * a.xs;
    const int w = a.w;
    const int tok = wv * 32 + l31;
    const int tokc = tok < w ? tok : w - 1;
    RA_T(0);

    // ---- the front vectors, own tokens, Wq (K buffer), then Wk (W1), Wv (W0): the Q projection starts when the first three have landed
    // (vector-memory operations return in order: a counted wait leaves exactly the Wk / Wv copies of THIS wave in flight); Q = LN-fold(Wq . x)
    ra_dma_vecs<6>(cv, a.vec, wv, lane);
    const int ntile = (w + 31) >> 5;
    const int nchunk = (ntile + 4) / 5;
    Tile q;
    {
        RawTile xr;
        ra_load_tile_async(xr, xrow + (long long)tokc * a.xs, hi);
        ra_dma(a.wgt, Ks, wv, nwv, lane);
        ra_dma(a.wgt + 1 * C * C, W1, wv, nwv, lane);
        ra_dma(a.wgt + 2 * C * C, W0, wv, nwv, lane);
        const int nkv = 2 * ((32 - wv + nwv - 1) / nwv);         // DMA instructions this wave issued for Wk and Wv
        ra_wait_vmcnt(nkv < 16 ? nkv : 0);
        Tile xa;
        ra_settle_tile(xa, xr);
        float mean, rstd;
        ra_ln_stats(xa, a.ln_eps, mean, rstd);
        ra_barrier_lds();                                        // Wq and the vectors of every wave's share are in LDS
        RA_T(1);
        float16_t acc[4];
        ra_linear<false>(acc, Ks, xa, l31, hi);
        ra_epilogue<S2M2_ACT_NONE, true>(q, acc, cv, cv + C, mean, rstd, hi);
        RA_T(2);
    }

    // ---- attention state: lane = query, O^T tiles of 32 channels
    float16_t oacc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};
    const float scale2 = a.scale * 1.44269504088896340736f;

#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        const int nt = ntile - 5 * c < 5 ? ntile - 5 * c : 5;    // key tiles of this chunk
        if (c == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); RA_T(6); }   // this wave's share of Wk / Wv has landed
        ra_barrier_lds();                                        // Q done with the K buffer / previous chunk's K, V consumed
        if (c < 2) RA_T(3 + 3 * c);
        // ---- projection: items [0, nt) = K tiles, [nt, 2 nt) = V tiles, dealt to the waves
#pragma unroll 1
        for (int item = wv; item < ((a.dbg & 4) ? 0 : 2 * nt); item += nwv) {
            const bool isv = item >= nt;
            const int tl = isv ? item - nt : item;               // tile within the chunk
            int st_tok = (5 * c + tl) * 32 + l31;
            st_tok = st_tok < w ? st_tok : w - 1;
            Tile sa;
            ra_load_tile(sa, srcrow + (long long)st_tok * a.xs, hi);
            float mean, rstd;
            ra_ln_stats(sa, a.ln_eps, mean, rstd);
            if (!isv) {
                half_t* kr = Ks + (tl * 32 + l31) * WRS + 8 * hi;
#pragma unroll 1
                for (int T = 0; T < 4; ++T) {
                    float16_t acc;
                    ra_linear_tile<false>(acc, W1, sa, T, l31, hi);
#pragma unroll
                    for (int hq = 0; hq < 2; ++hq) {              // fragments 2T, 2T + 1 of the K row, in the k order the Q fragments use
                        half8_t kv;
#pragma unroll
                        for (int gq = 0; gq < 2; ++gq) {
                            const int g = 2 * hq + gq, co = 32 * T + 8 * g;
                            float bv[4], ws[4];
                            ra_cvec(bv, cv + 2 * C, co, hi);
                            ra_cvec(ws, cv + 3 * C, co, hi);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                kv[4 * gq + e] = from_f32<half_t>(__builtin_fmaf(rstd, __builtin_fmaf(-mean, ws[e], acc[4 * g + e]), bv[e]));
                        }
                        *reinterpret_cast<half8_t*>(kr + 16 * (2 * T + hq)) = kv;
                    }
                }
            } else {
                // V^T: lane = channel, registers = tokens; the tokens' LayerNorm statistics travel through a wave-private LDS row
                if (hi == 0) stat[l31] = float2_t{mean, rstd};
                __builtin_amdgcn_wave_barrier();
#pragma unroll 1
                for (int T = 0; T < 4; ++T) {
                    float16_t acc;
                    ra_linear_tile<true>(acc, W0, sa, T, l31, hi);
                    const int d = 32 * T + l31;
                    const float bv = cv[4 * C + d], ws = cv[5 * C + d];
                    half_t* vr = Vt + d * VRS + tl * 32 + 8 * hi;
#pragma unroll
                    for (int sp = 0; sp < 2; ++sp) {
                        half8_t hv;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = 8 * sp + e;
                            const float2_t mr = stat[(r & 3) + 8 * (r >> 2) + 4 * hi];
                            hv[e] = from_f32<half_t>(__builtin_fmaf(mr.y, __builtin_fmaf(-mr.x, ws, acc[r]), bv));
                        }
                        *reinterpret_cast<half8_t*>(vr + 16 * sp) = hv;
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (c < 2) RA_T(4 + 3 * c);
        ra_barrier_lds();
        if (c < 2) RA_T(5 + 3 * c);
        if (c == nchunk - 1) {                                   // Wk / Wv and the front vectors are dead: proj -> W0, ffn.0 -> W1 and the tail's
            ra_dma(a.wgt + 3 * C * C, W0, wv, nwv, lane);        // vectors land under the attention below
            ra_dma(a.wgt + 4 * C * C, W1, wv, nwv, lane);
            ra_dma_vecs<6>(cv, a.vec + 6 * C, wv, lane);
        }
        // ---- attention of this wave's 32 queries over the chunk's keys (K4's per-32-key online softmax)
#pragma unroll 1
        for (int t = 0; t < ((a.dbg & 1) ? 0 : nt); ++t) {
            const int key0 = (5 * c + t) * 32;
            const bool full = key0 + 32 <= w;
#pragma unroll
            for (int hh = 0; hh < H; ++hh) {
                float16_t sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
                const half_t* kp = Ks + (t * 32 + l31) * WRS + 8 * hi;
                if (H == 1) {
#pragma unroll
                    for (int s = 0; s < 8; ++s) { Frag<half_t> kf; load_frag(kf, kp + 16 * s); mma32(sacc, kf, q[s]); }
                } else if (hh == 0) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) { Frag<half_t> kf; load_frag(kf, kp + 16 * s); mma32(sacc, kf, q[s]); }
                } else {
#pragma unroll
                    for (int s = 4; s < 8; ++s) { Frag<half_t> kf; load_frag(kf, kp + 16 * s); mma32(sacc, kf, q[s]); }
                }
                float smax = -INFINITY;
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) smax = fmaxf(smax, sacc[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        sacc[r] = key0 + acc_row(r, lane) < w ? sacc[r] : -INFINITY;
                        smax = fmaxf(smax, sacc[r]);
                    }
                }
                smax = fmaxf(smax, __shfl_xor(smax, 32, 64));
                const float mo = hh == 0 ? m_run[0] : m_run[1];
                const float m_new = fmaxf(mo, smax * scale2);
                const float alpha = __builtin_amdgcn_exp2f(mo - m_new);
                float p[16], lsum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[r] = (a.dbg & 2) ? sacc[r] : __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], scale2, -m_new));
                    lsum += p[r];
                }
                if (hh == 0) { l_run[0] = l_run[0] * alpha + lsum; m_run[0] = m_new; }
                else { l_run[1] = l_run[1] * alpha + lsum; m_run[1] = m_new; }
                const bool moved = __builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0;
                Frag<half_t> pf[2];
#pragma unroll
                for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pf[sp].v[e] = (half_t)p[8 * sp + e];
                // O^T tiles of this head: H = 1: all four; H = 2: tiles 2hh, 2hh + 1
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const bool mine = H == 1 || (dt >> 1) == hh;
                    if (mine) {
                        if (moved) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                        }
                        const half_t* vp = Vt + (dt * 32 + l31) * VRS + t * 32 + 8 * hi;
#pragma unroll
                        for (int sp = 0; sp < 2; ++sp) { Frag<half_t> vf; load_frag(vf, vp + 16 * sp); mma32(oacc[dt], vf, pf[sp]); }
                    }
                }
            }
        }
    }
    RA_T(9);
    __syncthreads();                                             // K / V are dead; proj / ffn.0 / the tail's vectors have landed (drains the DMA)
    RA_T(10);
    ra_dma(a.wgt + 5 * C * C, Ks, wv, nwv, lane);                // ffn.2 into the K buffer, under proj and ffn.0
    // ---- normalise: the attention output of this wave's tokens as the next layer's B operand
    Tile o;
    {
        const float i0 = 1.0f / (l_run[0] + __shfl_xor(l_run[0], 32, 64));
        const float i1 = H == 1 ? i0 : 1.0f / (l_run[1] + __shfl_xor(l_run[1], 32, 64));
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const float inv = dt < 2 ? i0 : i1;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[2 * dt + (r >> 3)].v[r & 7] = from_f32<half_t>(oacc[dt][r] * inv);
        }
    }
    Tile z;                                                      // the residual stream: the wave's own tokens again (L2)
    ra_load_tile(z, xrow + (long long)tokc * a.xs, hi);
    RA_T(11);
    float16_t acc[4];
    if (!(a.dbg & 8)) {
    ra_linear<false>(acc, W0, o, l31, hi);
    ra_epilogue<S2M2_ACT_NONE, false>(o, acc, cv, nullptr, 0.f, 0.f, hi);
    }
    ra_add(z, o);                                                // z' = z + proj(o)
    float mean, rstd;
    ra_ln_stats(z, a.ln_eps, mean, rstd);
    if (!(a.dbg & 8)) {
    ra_linear<false>(acc, W1, z, l31, hi);
    ra_epilogue<S2M2_ACT_GELU, true>(o, acc, cv + C, cv + 2 * C, mean, rstd, hi);
    }
    RA_T(12);
    __syncthreads();                                             // ffn.2 has landed (every wave's share)
    RA_T(13);
    if (!(a.dbg & 8)) {
    ra_linear<false>(acc, Ks, o, l31, hi);
    ra_epilogue<S2M2_ACT_NONE, false>(o, acc, cv + 3 * C, nullptr, 0.f, 0.f, hi);
    }
    ra_add(o, z);                                                // out = ffn(LN(z')) + z'
    RA_T(14);
    if (tok < w) ra_store_tile(a.out + ((long long)row * w + tok) * a.os, o, hi);
    RA_T(15);
    if (a.ln_out) {
        // LayerNorm with affine of the stored rows (DispInit's layer_norm, submodules.py:165,216): two passes in fp32, biased variance
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)o[k].v[e];
        s += __shfl_xor(s, 32);
        const float mu = s * (1.0f / (float)C);
        float sq = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)o[k].v[e] - mu; sq = __builtin_fmaf(d, d, sq); }
        sq += __shfl_xor(sq, 32);
        const float rs = rsqrtf(sq * (1.0f / (float)C) + a.ln_out_eps);
#pragma unroll
        for (int k = 0; k < 8; ++k)
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                float g[4], b[4];
                ra_cvec(g, cv + 4 * C, 16 * k + 8 * hq, hi);
                ra_cvec(b, cv + 5 * C, 16 * k + 8 * hq, hi);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    z[k].v[4 * hq + e] = from_f32<half_t>(__builtin_fmaf(((float)o[k].v[4 * hq + e] - mu) * rs, g[e], b[e]));
            }
        if (tok < w) ra_store_tile(a.ln_out + ((long long)row * w + tok) * a.ls, z, hi);
    }
}

}  // namespace s2m2

#if S2M2_RA_TRACE
extern "C" int s2m2_debug_ra_trace(void* host, size_t bytes, int clear) {
    void* dev = nullptr;
    if (hipGetSymbolAddress(&dev, HIP_SYMBOL(s2m2::g_ra_trace)) != hipSuccess) return 1;
    if (clear) return hipMemset(dev, 0, sizeof(s2m2::g_ra_trace)) == hipSuccess ? 0 : 1;
    return hipMemcpy(host, dev, bytes < sizeof(s2m2::g_ra_trace) ? bytes : sizeof(s2m2::g_ra_trace), hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

extern "C" int s2m2_row_attn_supported(int C, int heads, int w, int dtype) {
    return dtype == S2M2_F16 && C == 128 && (heads == 1 || heads == 2) && w >= 8 && w <= 32 * s2m2::ra::MAXW;
}

static int row_attn_impl(const s2m2_rowattn_desc* d, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(d, "row_attn: null descriptor");
    S2M2_REQUIRE(s2m2_row_attn_supported(d->C, d->heads, d->w, d->dtype), "row_attn: C=%d heads=%d w=%d dtype=%d (fp16, C = 128, 1 or 2 heads, 8 <= w <= 320)",
                 d->C, d->heads, d->w, d->dtype);
    S2M2_REQUIRE(d->x && d->out && d->x != d->out, "row_attn: x / out must be distinct non-null tensors (rows read their partner rows)");
    S2M2_REQUIRE(d->nimg > 0 && d->h > 0 && (long long)d->nimg * d->h < (1LL << 30), "row_attn: bad shape");
    S2M2_REQUIRE(!d->cross || d->nimg % 2 == 0, "row_attn: cross attention needs an even number of images (left | right halves of the batch)");
    S2M2_REQUIRE(d->x_stride >= 128 && d->x_stride % 4 == 0 && d->out_stride >= 128 && d->out_stride % 4 == 0, "row_attn: token strides must be >= 128 and multiples of 4");
    S2M2_REQUIRE(d->weights && d->vectors && (reinterpret_cast<uintptr_t>(d->weights) & 15) == 0 && (reinterpret_cast<uintptr_t>(d->vectors) & 15) == 0,
                 "row_attn: weights / vectors must be non-null and 16-byte aligned");
    S2M2_REQUIRE(d->ln_eps > 0.f, "row_attn: ln_eps must be positive");
    if (d->ln_out)
        S2M2_REQUIRE(d->ln_out_eps > 0.f && d->ln_out_stride >= 128 && d->ln_out_stride % 4 == 0,
                     "row_attn: ln_out needs a positive eps and a token stride >= 128 (multiple of 4)");
    RowAttnArgs a;
    a.x = static_cast<const half_t*>(d->x); a.out = static_cast<half_t*>(d->out); a.ln_out = static_cast<half_t*>(d->ln_out);
    a.xs = d->x_stride; a.os = d->out_stride; a.ls = d->ln_out_stride;
    a.nimg = d->nimg; a.h = d->h; a.w = d->w; a.heads = d->heads;
    a.src_off = d->cross ? (d->nimg / 2) * d->h : 0;
    a.xcd_rows = (d->xcd_hint && d->h % 8 == 0) ? d->h / 8 : 0;
    a.wgt = static_cast<const half_t*>(d->weights); a.vec = d->vectors;
    a.ln_eps = d->ln_eps; a.ln_out_eps = d->ln_out_eps;
    a.scale = 1.0f / sqrtf((float)(128 / d->heads));
    a.zero = nullptr;
    static const int dbg = getenv("S2M2_RA_DBG") ? atoi(getenv("S2M2_RA_DBG")) : 0;
    a.dbg = dbg;
    const int nwv = (d->w + 31) / 32;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned grid = (unsigned)(d->nimg * d->h);
    auto launch = [&](auto kern, size_t* granted) -> int {
        if (reserve_lds(reinterpret_cast<const void*>(kern), ra::LDS_BYTES, granted, "row_attn")) return 1;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * nwv), ra::LDS_BYTES, st, a);
        return check_launch("row_attn");
    };
    static size_t granted[4][kMaxDevices] = {};                      // per instantiation
    if (nwv <= 5) return d->heads == 1 ? launch(row_attn_kernel<320, 1>, granted[0]) : launch(row_attn_kernel<320, 2>, granted[1]);
    return d->heads == 1 ? launch(row_attn_kernel<640, 1>, granted[2]) : launch(row_attn_kernel<640, 2>, granted[3]);
}
extern "C" int s2m2_row_attn(const s2m2_rowattn_desc* d, void* stream) {
    return s2m2::plan_dispatch_desc<s2m2_rowattn_desc>("s2m2_row_attn", &row_attn_impl, d, stream);
}
