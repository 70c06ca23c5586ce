// K2 -- Sinkhorn optimal transport with dustbins + argmax + 5-tap window regression, one workgroup per image row.
//
// Replaces DispInit._optimal_transport/_sinkhorn (log-domain, ot_iter sweeps of logsumexp_stable), the probability recovery,
// argmax and the gather-based window expectation
// (/root/reference/src/s2m2/core/model/submodules.py:147-152,169-201,211-241; SURVEY.md A5+A6, Appendix A 3-10).
//
// For one (b,y) the score matrix S (w x w, + one dustbin row and column of zeros) enters
//     v_j = log nu_j - LSE_i(S_ij + u_i)      (column sweep; first one with u = 0)
//     u_i = log mu_i - LSE_j(S_ij + v_j)      (row sweep)
// ot_iter times, then P_ij = exp(S_ij + u_i + v_j + log 2w) for i,j < w.  With use_positivity the reference fills j > i with -1e4,
// which underflows to exactly 0 after exp in fp32, so those entries are skipped (bit-identical).
//
// The reference materialises ~35 full-volume temporaries (7 exp sweeps).  Here S is read ot_iter + 1 times, row-wise, with 16-byte
// loads: while a row is in registers its group of lanes computes u_i (exact max-then-sum log-sum-exp) AND immediately feeds
// S_ij + u_i into per-lane column accumulators for the next v -- the row sweep of iteration k and the column sweep of iteration
// k+1 are one pass; the last pass goes straight on to the probabilities, argmax (first maximum), row mass and window regression.
// u, v and the per-wave column partials live in LDS.
//
// Work mapping (round 2): a row belongs to a GROUP of GL = 16 / 32 / 64 lanes, a lane owns 8 consecutive columns in each of up to
// 3 column chunks of 8*GL columns, so a wave sweeps 64/GL rows at a time and the row reductions are DPP steps inside the group
// (no v_readlane round trips).  With one row per wave (round 1) a 304-column row used 38 of 64 lanes and, under the positivity
// triangle, 19 on average.  The column accumulators are "lazy" log-sum-exp states: the stabiliser only moves when an element
// exceeds it by more than 40 (checked once per 8 elements with a wave vote), otherwise an element costs a subtract, an exp and an
// add -- the exact sum of exp(x - m) for a fixed m, just not the tightest m.
//
// Round 6: the kernel is VALU-issue bound (rocprofv3 counters, profiles/r06/pmc_k2_sq_lds.txt: one block of 8 waves per CU, VALU busy
// 0.37 per wave = 0.74 per SIMD, ~15 instructions per element and pass), so the sweeps were rebuilt around the instruction count:
//   * everything lives in the log2 domain (u, v, the marginals and the stabilisers are the natural-log quantities times log2 e):
//     x = S * log2e + v is ONE fused multiply-add straight from the fp16 element (v_fma_mix_f32), exp2 is the bare v_exp_f32 -- the
//     convert, the add and the multiply in front of every exp are gone.  Algebraically the same algorithm on S * (log2e rounded to
//     fp32), i.e. on inputs perturbed by 1.4e-8 relative;
//   * [TRI] the LDS copy of the triangle is stored MASKED (entries right of the diagonal = -inf inside the diagonal piece), holds the
//     dustbin row (zeros) as row w, and has one all -inf piece that every read right of the diagonal / below the dustbin row is pointed
//     at: the fused passes decode without a single compare or select, and no load sits under an exec mask;
//   * v is padded to whole chunks: its 16-byte reads are unconditional;
//   * the row loop is three loops with a compile-time number of live chunks (1, 2, .. NCH; under the triangle the early rows need
//     fewer): straight-line code, no per-chunk branches;
//   * subtract / row sum / column multiply-add work on register pairs (v_pk_add_f32, v_pk_fma_f32), the 16-lane reductions are
//     v_max_f32_dpp / v_add_f32_dpp / v_min_i32_dpp (one instruction per butterfly step).
#include "common.h"
#include "plan.h"
#include <stdlib.h>

namespace s2m2 {

constexpr float kLazy = 40.0f;          // 2^40 * (columns) stays far below the fp32 range
constexpr float kNegBig = -1.0e30f;     // "no element yet" stabiliser (finite: -inf - -inf would be NaN)
constexpr float kL2E = 1.44269504088896340736f;

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }      // v_exp_f32 (arguments <= ~0 here: no denormal pre-scaling)
__device__ __forceinline__ float lg2(float x) { return __builtin_amdgcn_logf(x); }       // v_log_f32 (arguments >= 1e-30: normal numbers)

struct LSE {                            // running log2-sum-exp2 state: z = sum of 2^(x - m)
    float m, z;
    __device__ __forceinline__ void init() { m = kNegBig; z = 0.f; }
    __device__ __forceinline__ void add_exact(float x) {       // online update with the maximum as stabiliser (x may be -inf: no-op)
        if (x > m) { z = z * ex2(m - x) + 1.0f; m = x; }
        else z += ex2(x - m);
    }
    __device__ __forceinline__ void merge(float m2, float z2) {
        const float mn = fmaxf(m, m2);
        z = z * ex2(m - mn) + z2 * ex2(m2 - mn);
        m = mn;
    }
    // logsumexp_stable: m + log(max(sum, 1e-30))   (log2 domain: the same sum, the same clamp)
    __device__ __forceinline__ float value() const { return m + lg2(fmaxf(z, 1e-30f)); }
};

// butterfly steps inside aligned groups of 16 lanes as single DPP VALU instructions: quad_perm[1,0,3,2], quad_perm[2,3,0,1],
// row_half_mirror, row_mirror.  (The s_nop covers the "VALU write -> DPP read" wait states, which nobody inserts inside inline asm.)
#define S2M2_DPP16(NAME, TYPE, OP)                                                                                                   \
    __device__ __forceinline__ TYPE NAME(TYPE x) {                                                                                   \
        TYPE a, b, c, d;                                                                                                             \
        asm("s_nop 1\n\t" OP " %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(a) : "v"(x));                       \
        asm("s_nop 1\n\t" OP " %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(b) : "v"(a));                       \
        asm("s_nop 1\n\t" OP " %0, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "=v"(c) : "v"(b));                           \
        asm("s_nop 1\n\t" OP " %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(c));                                \
        return d;                                                                                                                    \
    }
S2M2_DPP16(row16_max_f, float, "v_max_f32_dpp")
S2M2_DPP16(row16_sum_f, float, "v_add_f32_dpp")
S2M2_DPP16(row16_min_i, int, "v_min_i32_dpp")
#undef S2M2_DPP16
// reductions over aligned groups of GL lanes, result in every lane of the group: 4 DPP steps cover 16 lanes, ds_bpermute the rest
template <int GL> __device__ __forceinline__ float group_sum_f(float x) {
    x = row16_sum_f(x);
    if (GL >= 32) x += __shfl_xor(x, 16, 64);
    if (GL >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}
template <int GL> __device__ __forceinline__ float group_max_f(float x) {
    x = row16_max_f(x);
    if (GL >= 32) x = fmaxf(x, __shfl_xor(x, 16, 64));
    if (GL >= 64) x = fmaxf(x, __shfl_xor(x, 32, 64));
    return x;
}
template <int GL> __device__ __forceinline__ int group_min_i(int x) {
    x = row16_min_i(x);
    if (GL >= 32) x = min(x, __shfl_xor(x, 16, 64));
    if (GL >= 64) x = min(x, __shfl_xor(x, 32, 64));
    return x;
}

// 16-byte pieces in front of row i of the LDS-resident triangle (row r holds its columns 0 .. r, rounded up to whole pieces)
__host__ __device__ __forceinline__ int tri_pieces(int i, int ppe) {        // ppe: elements per piece
    const int q = i / ppe, rem = i - q * ppe;
    return i + ppe * ((q * (q - 1)) >> 1) + q * rem;               // sum_{r < i} (floor(r / ppe) + 1)
}

// LDS layout of one block: u [ns] | pm [NWV][ns] | pz [NWV][ns] | flags [4] | v [nvs] | (TRI) the masked triangle, the dustbin row, the -inf piece
template <int NWV, int GL, int NCH>
struct K2Lds {
    __host__ __device__ static int ns(int w) { return (w + 4) & ~3; }                   // w + 1 entries, 16-byte rows
    __host__ __device__ static int nvs(int w) { return ns(w) > NCH * 8 * GL + 8 ? ns(w) : NCH * 8 * GL + 8; }     // v: whole chunks (unconditional 16-byte reads)
    __host__ __device__ static size_t vec_bytes(int w) { return ((size_t)(1 + 2 * NWV) * ns(w) + 4 + nvs(w)) * sizeof(float); }
    __host__ __device__ static size_t tri_pieces_total(int w, int vec) { return (size_t)tri_pieces(w, vec) + w / vec + 1; }
};

template <typename TI> __device__ __forceinline__ raw16_t pack_piece(const float* x);
template <> __device__ __forceinline__ raw16_t pack_piece<half_t>(const float* x) {   // exact: the values came from fp16 (or are -inf / 0)
    raw16_t r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __builtin_bit_cast(float, __builtin_amdgcn_cvt_pkrtz(x[2 * k], x[2 * k + 1]));
    return r;
}
template <> __device__ __forceinline__ raw16_t pack_piece<float>(const float* x) { return raw16_t{x[0], x[1], x[2], x[3]}; }

template <int N> struct IC { static constexpr int value = N; };
// f(IC<J0>), f(IC<J0 + 1>), ... f(IC<J1 - 1>)
template <int J0, int J1, typename F> __device__ __forceinline__ void static_range(F&& f) {
    if constexpr (J0 < J1) { f(IC<J0>{}); static_range<J0 + 1, J1>(f); }
}

// TRI: the block keeps the row's lower cost-volume triangle (use_positivity: j <= i) in LDS -- pass 0 copies the pieces it reads from
// global memory, the ot_iter later sweeps read LDS: the volume is read from HBM / MALL ONCE (the algorithmic minimum) and the latency
// of a row fetch drops from a memory round trip to an LDS read.  Needs (w / 8 + 1) * w / 2 * 16 bytes: w <= ~380 for fp16.
template <typename TI, int NWV, int GL, int NCH, bool TRI = false>
__global__ __launch_bounds__(NWV * 64) void sinkhorn_regress_kernel(const TI* __restrict__ cv, float* __restrict__ disp,
                                                                    float* __restrict__ conf, float* __restrict__ occ,
                                                                    int32_t* __restrict__ amax, int w, int ot_iter, int use_pos, int pitch) {
    constexpr int VEC = 16 / sizeof(TI);
    constexpr int PPC = 8 / VEC;                           // 16-byte pieces per lane and chunk (8 columns)
    constexpr int CW = 8 * GL;                             // columns per chunk
    constexpr int RPW = 64 / GL;                           // rows per wave and step
    constexpr int RPB = RPW * NWV;                         // rows per block and step
    constexpr int NE = 8 * NCH;                            // columns per lane
    using L = K2Lds<NWV, GL, NCH>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = w + 1;                                   // padded size
    const int ns = L::ns(w);                               // row stride of the LDS vectors (16-byte aligned rows)
    float* u = reinterpret_cast<float*>(smem);             // [ns]                               (log2 domain, like v)
    float* pm = u + ns;                                    // [NWV][ns]  per-wave column partial stabiliser
    float* pz = pm + NWV * ns;                             // [NWV][ns]  per-wave column partial sum
    int* flags = reinterpret_cast<int*>(pz + NWV * ns);    // [4]
    float* v = reinterpret_cast<float*>(flags + 4);        // [nvs]
    // [TRI] masked lower triangle behind the vectors, 16-byte aligned; row w = the dustbin row (zeros); then one piece of -inf
    raw16_t* tri = reinterpret_cast<raw16_t*>(smem + ((L::vec_bytes(w) + 15) & ~(size_t)15));
    const int ninf_idx = tri_pieces(w, VEC) + w / VEC;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = lane / GL, pl = lane % GL;             // row group inside the wave, position inside the group
    const TI* S = cv + (size_t)blockIdx.x * w * pitch;          // volume rows `pitch` elements apart (>= w, multiple of 8)
    const float log_row = -lg2(2.0f * w);                  // log2(1/(2w))   marginal of a regular row/column
    const float log_bin = -1.0f;                           // log2(w/(2w))   marginal of the dustbin
    const float log2w = lg2(2.0f * w);
    float* od = disp + (size_t)blockIdx.x * w;
    float* oc = conf + (size_t)blockIdx.x * w;
    float* oo = occ + (size_t)blockIdx.x * w;

    // row i (i == w: the dustbin row, S = 0, never masked): raw 16-byte pieces of the lane's columns j = c*CW + pl*8 + 0..7 from global memory
    auto fetch_global = [&](int i, raw16_t (&raw)[NCH][PPC]) __attribute__((always_inline)) {
        const TI* Si = S + (size_t)(i < w ? i : 0) * pitch;
        const int jend = i < w ? (use_pos ? i + 1 : w) : 0;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < PPC; ++q) {
                const int j0 = c * CW + pl * 8 + q * VEC;
                if (j0 < jend) raw[c][q] = global_load16(Si + j0);          // w % 8 == 0: a piece starting inside [0, w) is whole
            }
    };
    // [TRI] the same pieces from the masked LDS triangle: rows 0 .. w hold what they hold, everything else is the -inf piece
    auto fetch_tri = [&](int i, raw16_t (&raw)[NCH][PPC], int nc) __attribute__((always_inline)) {
        const int jl = i < w ? i + 1 : (i == w ? w : 0);
        const int base = tri_pieces(i < w ? i : w, VEC) + pl * PPC;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < PPC; ++q) {
                if (c >= nc) continue;
                const int j0 = c * CW + pl * 8 + q * VEC;
                raw[c][q] = tri[j0 < jl ? base + c * (CW / VEC) + q : ninf_idx];
            }
    };
    auto fetch_row = [&](int i, raw16_t (&raw)[NCH][PPC]) __attribute__((always_inline)) {
        if constexpr (TRI) fetch_tri(i, raw, NCH);
        else if (i <= w) fetch_global(i, raw);
    };
    // raw pieces of chunk c of row i -> natural-domain floats: masked triangle / past the row = -inf, the dustbin row = 0.
    // fast: every column of the chunk is valid for EVERY row of this wave in this step (regular rows, nothing masked): a plain convert.
    // The per-element compare / select (two instructions per element) is left to the one or two chunks the diagonal crosses.
    auto decode_chunk = [&](int c, int i, const raw16_t (&raw)[PPC], float (&x)[8], bool fast) __attribute__((always_inline)) {
        const int jend = i < w ? (use_pos ? i + 1 : w) : (i == w ? w : 0);
        if (fast) {
#pragma unroll
            for (int q = 0; q < PPC; ++q) {
                const Vec16<TI> r = __builtin_bit_cast(Vec16<TI>, raw[q]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) x[q * VEC + e] = to_f32(r.v[e]);
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < PPC; ++q) {
            const Vec16<TI> r = __builtin_bit_cast(Vec16<TI>, raw[q]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int j = c * CW + pl * 8 + q * VEC + e;
                x[q * VEC + e] = j < jend ? (i < w ? to_f32(r.v[e]) : 0.f) : -INFINITY;
            }
        }
    };
    // chunks fully valid for every row of this wave in the step starting at row i0 (uniform per wave)
    auto chunks_fast = [&](int i0) {
        const int imin = i0 + wv * RPW, imax = imin + RPW - 1;
        if (imax >= w) return 0;                                  // a dustbin / idle row in the wave: general decode
        return (use_pos ? imin + 1 : w) / CW;
    };

    const int r0 = wv * RPW + grp;                         // rows of this group: r0, r0 + RPB, ...
    // chunks that hold unmasked columns for ANY row of this wave in the step starting at row i0 (uniform per wave): under the
    // positivity triangle row i ends at column i, so the early steps skip the right-hand chunks altogether
    auto chunks_needed = [&](int i0) {
        const int imax = i0 + wv * RPW + RPW - 1;
        return (!use_pos || imax >= w) ? NCH : min(NCH, imax / CW + 1);
    };
    // fast column sweep lost a column (underflow): redo the pass exactly.  Three flags used in turn: pass p raises flags[p % 3] and
    // clears flags[(p + 1) % 3] for its successor, so a wave still reading the flag of pass p - 1 never sees it reset
    if (tid < 3) flags[tid] = 0;                           // (ordered before the first use by the barriers of pass 0)
    // the padding of v (entries n .. nvs-1) is never written by the sweeps but IS read by the unconditional 16-byte reads: it is added
    // to masked (-inf) columns only, and must be finite for that (whatever the previous kernel left in LDS may be +inf / NaN)
    for (int j = n + tid; j < L::nvs(w); j += NWV * 64) v[j] = 0.f;
    if (tid < ns - n) u[n + tid] = 0.f;
    if constexpr (TRI) {
        if (tid == 0) {
            const float ninf = __builtin_bit_cast(float, sizeof(TI) == 2 ? 0xFC00FC00u : 0xFF800000u);
            tri[ninf_idx] = raw16_t{ninf, ninf, ninf, ninf};
        }
    }

    // ---- exact column sweep: v_j = log nu_j - LSE_i(S_ij + u_i) with lazy-maximum accumulators (pass 0, where u = 0, and the
    // fallback of the fast sweep below)
    auto exact_cols = [&](bool use_u) __attribute__((always_inline)) {        // (use_u false = pass 0: the one sweep that reads global memory in a TRI block)
        float2_t cm[NE / 2], cz[NE / 2];                   // per column: stabiliser and sum of 2^(t - stabiliser)
        LSE bin;
#pragma unroll
        for (int c = 0; c < NE / 2; ++c) { cm[c] = float2_t{kNegBig, kNegBig}; cz[c] = float2_t{0.f, 0.f}; }
        bin.init();
        for (int i0 = 0; i0 <= w; i0 += RPB) {
            const int i = i0 + r0;
            const bool active = i <= w;
            const int nchw = chunks_needed(i0);
            raw16_t raw[NCH][PPC];
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int q = 0; q < PPC; ++q) raw[c][q] = (raw16_t){0.f, 0.f, 0.f, 0.f};
            const bool from_tri = TRI && use_u;
            if (from_tri) fetch_tri(i, raw, nchw);                      // (masked copy: plain converts)
            else if (active) fetch_global(i, raw);
            const int nfast = chunks_fast(i0);
            const int idec = active ? i : w + 1;
            const float ua = active ? (use_u ? u[i] : 0.f) : -INFINITY;
            raw16_t* trow = tri;
            if constexpr (TRI) trow = tri + tri_pieces(i < w ? i : w, VEC) + pl * PPC;
            const int jl = i < w ? i + 1 : w;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c >= nchw) continue;                          // (wave-uniform)
                float x[8];
                decode_chunk(c, idec, raw[c], x, from_tri || c < nfast);
                if constexpr (TRI) {
                    if (!use_u && active) {                       // pass 0: the masked copy of this row (the dustbin row: zeros) for the later passes
#pragma unroll
                        for (int q = 0; q < PPC; ++q)
                            if (c * CW + pl * 8 + q * VEC < jl) trow[c * (CW / VEC) + q] = pack_piece<TI>(&x[q * VEC]);
                    }
                }
                float2_t t[4], d[4];
                float dmax = -INFINITY;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    t[k] = float2_t{__builtin_fmaf(x[2 * k], kL2E, ua), __builtin_fmaf(x[2 * k + 1], kL2E, ua)};
                    d[k] = t[k] - cm[c * 4 + k];
                    dmax = fmaxf(dmax, fmaxf(d[k][0], d[k][1]));
                }
                if (__builtin_amdgcn_ballot_w64(dmax > kLazy)) {                  // rare: a stabiliser is too far below its new element -> it moves up to it
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            if (t[k][h] > cm[c * 4 + k][h]) { cz[c * 4 + k][h] *= ex2(cm[c * 4 + k][h] - t[k][h]); cm[c * 4 + k][h] = t[k][h]; }
#pragma unroll
                    for (int k = 0; k < 4; ++k) d[k] = t[k] - cm[c * 4 + k];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) cz[c * 4 + k] += float2_t{ex2(d[k][0]), ex2(d[k][1])};
            }
            bin.add_exact(ua);
        }
        // combine: across the row groups of a wave (lanes with equal pl), then across waves through LDS
#pragma unroll
        for (int off = GL; off < 64; off <<= 1) {
#pragma unroll
            for (int c = 0; c < NE; ++c) {
                LSE t{cm[c / 2][c & 1], cz[c / 2][c & 1]};
                t.merge(__shfl_xor(t.m, off, 64), __shfl_xor(t.z, off, 64));
                cm[c / 2][c & 1] = t.m; cz[c / 2][c & 1] = t.z;
            }
            bin.merge(__shfl_xor(bin.m, off, 64), __shfl_xor(bin.z, off, 64));
        }
        __syncthreads();                                   // readers of pm / pz / v of the previous pass are done
        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < NE; ++c) {
                const int j = (c / 8) * CW + pl * 8 + (c % 8);
                if (j < w) { pm[wv * ns + j] = cm[c / 2][c & 1]; pz[wv * ns + j] = cz[c / 2][c & 1]; }
            }
            if (pl == 0) { pm[wv * ns + w] = bin.m; pz[wv * ns + w] = bin.z; }
        }
        __syncthreads();
        for (int j = tid; j < n; j += NWV * 64) {
            LSE t; t.init();
#pragma unroll
            for (int k = 0; k < NWV; ++k) t.merge(pm[k * ns + j], pz[k * ns + j]);
            v[j] = (j == w ? log_bin : log_row) - t.value();
        }
        __syncthreads();
    };

    exact_cols(false);                                     // pass 0: v = log nu - LSE_i(S)

    // ---- passes 1 .. ot_iter: row sweep (exact max-then-sum) fused with the next column sweep.  With e_ij = 2^(S_ij + v_j - m_i)
    // from the row sweep, the column term is 2^(S_ij + u_i - (c0 - v_j)) = e_ij * 2^(u_i + m_i - c0): ONE multiply-add per element
    // against the per-row factor f_i -- the stabiliser c0 - v_j is the column's previous log-sum-exp up to a constant, and
    // f_i <= 1 for c0 = log(1/2) (u_i + m_i <= log mu_i), so nothing can overflow; a column whose sum underflows (it would need a
    // dynamic range of e^80 inside the volume) raises a flag and the pass is redone with exact_cols.
    const float c0 = log_bin;
    for (int pass = 1; pass <= ot_iter; ++pass) {
        const bool last = pass == ot_iter;
        float2_t zc[NE / 2];
        float zbin = 0.f;
#pragma unroll
        for (int c = 0; c < NE / 2; ++c) zc[c] = float2_t{0.f, 0.f};
        const float vbin = v[w];
        int* flag = flags + pass % 3;
        if (tid == 0) flags[(pass + 1) % 3] = 0;
        raw16_t rcur[NCH][PPC], rnext[NCH][PPC];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < PPC; ++q) { rcur[c][q] = (raw16_t){0.f, 0.f, 0.f, 0.f}; rnext[c][q] = rcur[c][q]; }
        if constexpr (!TRI) fetch_row(r0, rcur);

        // one step = RPW rows per wave with NC live chunks (compile time: straight-line code)
        auto row_step = [&](auto nc_tag, auto last_tag, int i0) __attribute__((always_inline)) {
            constexpr int NC = decltype(nc_tag)::value;
            constexpr bool LAST = decltype(last_tag)::value != 0;
            const int i = i0 + r0;
            const bool active = i <= w;
            if constexpr (TRI) fetch_tri(i, rcur, NC);              // (LDS: no prefetch distance needed -- and 4 NCH registers fewer)
            else if (i + RPB <= w) fetch_row(i + RPB, rnext);     // global memory: the next row is in flight under this row's arithmetic
            // ---- x = S * log2e + v: the row sweep's exponent, dustbin column apart (S = 0)
            float2_t x[NC * 4];
            float mloc = -INFINITY;
            if constexpr (TRI) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float* vp = v + c * CW + pl * 8;
                    const float4_t va = *reinterpret_cast<const float4_t*>(vp), vb = *reinterpret_cast<const float4_t*>(vp + 4);
                    const float vv[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
                    for (int q = 0; q < PPC; ++q) {
                        const Vec16<TI> r = __builtin_bit_cast(Vec16<TI>, rcur[c][q]);
#pragma unroll
                        for (int e = 0; e < VEC; e += 2) {
                            float2_t t;
                            t[0] = __builtin_fmaf(to_f32(r.v[e]), kL2E, vv[q * VEC + e]);
                            t[1] = __builtin_fmaf(to_f32(r.v[e + 1]), kL2E, vv[q * VEC + e + 1]);
                            x[c * 4 + (q * VEC + e) / 2] = t;
                            mloc = fmaxf(mloc, fmaxf(t[0], t[1]));
                        }
                    }
                }
            } else {
                const int nfast = chunks_fast(i0);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    float s[8];
                    decode_chunk(c, active ? i : w + 1, rcur[c], s, c < nfast);
                    const float* vp = v + c * CW + pl * 8;
                    const float4_t va = *reinterpret_cast<const float4_t*>(vp), vb = *reinterpret_cast<const float4_t*>(vp + 4);
                    const float vv[8] = {va[0], va[1], va[2], va[3], vb[0], vb[1], vb[2], vb[3]};
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        float2_t t;
                        t[0] = __builtin_fmaf(s[e], kL2E, vv[e]);
                        t[1] = __builtin_fmaf(s[e + 1], kL2E, vv[e + 1]);
                        x[c * 4 + e / 2] = t;
                        mloc = fmaxf(mloc, fmaxf(t[0], t[1]));
                    }
                }
            }
            if constexpr (!TRI) {
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int q = 0; q < PPC; ++q) rcur[c][q] = rnext[c][q];
            }
            // ---- row sweep: u_i = log mu_i - LSE_j(S_ij + v_j), dustbin column included
            const float gmax = group_max_f<GL>(mloc);             // (-inf for an idle row: vbin takes over)
            const float m = fmaxf(gmax, vbin);
            const float2_t m2 = float2_t{m, m};
            float2_t s2 = float2_t{0.f, 0.f};
#pragma unroll
            for (int k = 0; k < NC * 4; ++k) {
                const float2_t d = x[k] - m2;
                x[k] = float2_t{ex2(d[0]), ex2(d[1])};                            // x <- e_ij
                s2 += x[k];
            }
            const float srow = group_sum_f<GL>(s2[0] + s2[1]);
            const float ebin = ex2(vbin - m);
            const float ui = (i == w ? log_bin : log_row) - (m + lg2(fmaxf(srow + ebin, 1e-30f)));
            if constexpr (!LAST) {
                if (pl == 0 && active) u[i] = ui;                                 // (only the exact fallback reads it)
                const float f = active ? ex2(ui + m - c0) : 0.f;
                const float2_t f2 = float2_t{f, f};
#pragma unroll
                for (int k = 0; k < NC * 4; ++k) zc[k] = x[k] * f2 + zc[k];
                zbin = __builtin_fmaf(ebin, f, zbin);
            } else if (i < w) {
                // ---- probabilities P_ij = e_ij * 2^(m_i + u_i + log 2w), argmax (first max wins), window regression, row mass
                const float ci = ui + log2w;
                const float gsc = ex2(m + ci);
                const int jend = use_pos ? i + 1 : w;
                // the maximum of the row is the element whose exponent was the row maximum: e == 2^(gmax - m) bit for bit (the same instruction
                // on the same operands).  Masked entries are 0 and column 0 is never masked, so they cannot win; scanning the lane's elements
                // backwards leaves its FIRST match.  (P = e * gsc is monotone in e: the argmax of the plan up to ties below one ulp.)
                const float emax = ex2(gmax - m);
                int bk = 0x7fffffff;
#pragma unroll
                for (int k = NC * 8 - 1; k >= 0; --k) bk = x[k / 2][k & 1] == emax ? k : bk;
                int bj = bk == 0x7fffffff ? bk : (bk >> 3) * CW + pl * 8 + (bk & 7);
                bj = group_min_i<GL>(bj);
                const float mass = srow * gsc;
                // 5 taps around the argmax, evaluated by lanes 0..4 of the group (zero outside [0,w) and in the masked triangle)
                const TI* Si = TRI ? reinterpret_cast<const TI*>(tri + tri_pieces(i, VEC)) : S + (size_t)i * pitch;
                const int jj = bj + pl - 2;
                float pk = 0.f;
                if (pl < 5 && jj >= 0 && jj < jend) pk = ex2(__builtin_fmaf(to_f32(Si[jj]), kL2E, ci + v[jj]));
                float cf = 0.f, num = 0.f;
#pragma unroll
                for (int k = 0; k < 5; ++k) {                                     // same summation order as the reference loop
                    const float pr = __shfl(pk, grp * GL + k, 64);
                    cf += pr;
                    num += pr * (float)(bj + k - 2);
                }
                if (pl == 0) {
                    const float corr = (num + 1e-4f) / (cf + 1e-4f);
                    od[i] = (float)i - corr;
                    oc[i] = cf;
                    oo[i] = mass;
                    if (amax) amax[(size_t)blockIdx.x * w + i] = bj;
                }
            }
        };
        {
            int i0 = 0;                                    // uniform trip count for every wave (groups past row w idle)
            if (!last) {
                static_range<1, NCH>([&](auto nc) __attribute__((always_inline)) {
                    for (; i0 <= w && chunks_needed(i0) <= decltype(nc)::value; i0 += RPB) row_step(nc, IC<0>{}, i0);
                });
                for (; i0 <= w; i0 += RPB) row_step(IC<NCH>{}, IC<0>{}, i0);
            } else {
                static_range<1, NCH>([&](auto nc) __attribute__((always_inline)) {
                    for (; i0 <= w && chunks_needed(i0) <= decltype(nc)::value; i0 += RPB) row_step(nc, IC<1>{}, i0);
                });
                for (; i0 <= w; i0 += RPB) row_step(IC<NCH>{}, IC<1>{}, i0);
            }
        }
        if (last) break;
        // ---- combine the partial column sums: across the row groups of a wave (lanes with equal pl), then across waves through LDS
#pragma unroll
        for (int off = GL; off < 64; off <<= 1) {
#pragma unroll
            for (int c = 0; c < NE / 2; ++c) {
                zc[c][0] += __shfl_xor(zc[c][0], off, 64);
                zc[c][1] += __shfl_xor(zc[c][1], off, 64);
            }
            zbin += __shfl_xor(zbin, off, 64);
        }
        __syncthreads();                                   // every row sweep of this pass has read v
        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < NE; ++c) {
                const int j = (c / 8) * CW + pl * 8 + (c % 8);
                if (j < w) pz[wv * ns + j] = zc[c / 2][c & 1];
            }
            if (pl == 0) pz[wv * ns + w] = zbin;
        }
        __syncthreads();
        for (int j = tid; j < n; j += NWV * 64) {
            float z = 0.f;
#pragma unroll
            for (int k = 0; k < NWV; ++k) z += pz[k * ns + j];
            if (!(z > 1e-30f) || !(z < 3.0e38f)) *flag = 1;                       // underflow (or NaN): this pass needs the exact sweep
            v[j] = (j == w ? log_bin : log_row) - ((c0 - v[j]) + lg2(z));
        }
        __syncthreads();
        if (*flag) exact_cols(true);                       // uniform for the block; rewrites every v_j from S and u
    }
}

template <typename TI, int GL, int NCH, bool TRI = false>
static int launch_sinkhorn(const void* cv, float* disp, float* conf, float* occ, int32_t* amax, int rows, int w, int ot_iter,
                           int use_pos, int pitch, hipStream_t st) {
    // 16 waves per row block where a lane's state (8 * NCH columns: values + two accumulator words each) fits 128 registers, else 8
    constexpr int NWV = NCH == 1 ? 16 : 8;
    using L = K2Lds<NWV, GL, NCH>;
    auto kern = sinkhorn_regress_kernel<TI, NWV, GL, NCH, TRI>;
    size_t lds = L::vec_bytes(w);
    if (TRI) {
        constexpr int VEC = 16 / sizeof(TI);
        const size_t tri_bytes = ((lds + 15) & ~(size_t)15) + L::tri_pieces_total(w, VEC) * 16;
        static const bool off = getenv("S2M2_K2_TRI") != nullptr && atoi(getenv("S2M2_K2_TRI")) == 0;   // A/B switch
        if (!use_pos || off || tri_bytes > 160 * 1024)
            return launch_sinkhorn<TI, GL, NCH, false>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, pitch, st);
        lds = tri_bytes;
    }
    if (lds > 160 * 1024) return set_error("sinkhorn: w=%d needs %zu bytes of LDS (at most about w = 1200)", w, lds);
    static size_t lds_granted[kMaxDevices] = {};                     // per instantiation
    if (reserve_lds(reinterpret_cast<const void*>(kern), lds, lds_granted, "sinkhorn")) return 1;
    hipLaunchKernelGGL(kern, dim3(rows), dim3(NWV * 64), lds, st, static_cast<const TI*>(cv), disp, conf, occ, amax, w, ot_iter, use_pos, pitch);
    return check_launch("sinkhorn_regress");
}

// lanes per row so that a row needs at most 3 chunks of 8 columns per lane: 16 lanes up to w = 384, 32 up to 768, 64 up to 1536.
// (r06, measured and dropped: 8 lanes per row with up to five 64-column chunks for w <= 320 -- the chunk is the granularity at which the masked
// part of a row is skipped, 184 instead of 223 columns swept per row at w = 304, and a step covers 8 rows per wave: 46.08 vs 46.02 us at c3,
// 23.9 vs 24.4 at c2, for ten more instantiations.)
template <typename TI>
static int dispatch_ppl(const void* cv, float* disp, float* conf, float* occ, int32_t* amax, int rows, int w, int ot_iter,
                        int use_pos, int pitch, hipStream_t st) {
#define S2M2_K2(GL)                                                                                                          \
    {                                                                                                                        \
        const int nch = (w + 8 * GL - 1) / (8 * GL);                                                                         \
        if (nch <= 1) return launch_sinkhorn<TI, GL, 1, GL == 16>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, pitch, st);          \
        if (nch <= 2) return launch_sinkhorn<TI, GL, 2, GL == 16>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, pitch, st);          \
        if (nch <= 3) return launch_sinkhorn<TI, GL, 3, GL == 16>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, pitch, st);          \
    }
    // (measured at w = 304: 32 lanes per row x 16 waves, 91.7 us, is no faster than 16 lanes x 8 waves, 90.4 us -- the row chain, not
    // the number of resident waves, sets the pace)
    if (w <= 384) S2M2_K2(16)
    if (w <= 768) S2M2_K2(32)
    if (w <= 1536) S2M2_K2(64)
#undef S2M2_K2
    return set_error("sinkhorn: w=%d too large (max 1536)", w);
}

}  // namespace s2m2

extern "C" size_t s2m2_sinkhorn_workspace_bytes(int B, int h, int w, int cv_dtype) {
    (void)B; (void)h; (void)w; (void)cv_dtype;
    return 0;                                    // u, v and the partials live in LDS
}

static int sinkhorn_regress_impl(const void* cv, float* disp, float* conf, float* occ, int32_t* argmax, int B, int h, int w,
                                     int ot_iter, int use_positivity, int cv_dtype, int cv_pitch, void* workspace, void* stream) {
    using namespace s2m2;
    (void)workspace;
    S2M2_REQUIRE(cv && disp && conf && occ, "sinkhorn: null pointer");
    S2M2_REQUIRE(B > 0 && h > 0 && w > 0 && ot_iter >= 1, "sinkhorn: bad arguments B=%d h=%d w=%d ot_iter=%d", B, h, w, ot_iter);
    S2M2_REQUIRE(w % 8 == 0, "sinkhorn: w=%d must be a multiple of 8 (image width multiple of 32)", w);
    if (cv_pitch == 0) cv_pitch = w;
    S2M2_REQUIRE(cv_pitch >= w && cv_pitch % 8 == 0, "sinkhorn: cv_pitch=%d must be a multiple of 8 and at least w=%d", cv_pitch, w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cv_dtype == S2M2_F16) return dispatch_ppl<half_t>(cv, disp, conf, occ, argmax, B * h, w, ot_iter, use_positivity, cv_pitch, st);
    if (cv_dtype == S2M2_F32) return dispatch_ppl<float>(cv, disp, conf, occ, argmax, B * h, w, ot_iter, use_positivity, cv_pitch, st);
    return set_error("sinkhorn: unsupported cv dtype %d", cv_dtype);
}
extern "C" int s2m2_sinkhorn_regress(const void* cv, float* disp, float* conf, float* occ, int32_t* argmax, int B, int h, int w,
                                     int ot_iter, int use_positivity, int cv_dtype, int cv_pitch, void* workspace, void* stream) {
    return s2m2::plan_dispatch("s2m2_sinkhorn_regress", &sinkhorn_regress_impl, stream, cv, disp, conf, occ, argmax, B, h, w, ot_iter, use_positivity, cv_dtype, cv_pitch, workspace);
}

