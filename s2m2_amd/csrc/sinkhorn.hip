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
#include "common.h"
#include "plan.h"
#include <stdlib.h>

namespace s2m2 {

constexpr float kLazy = 40.0f;          // exp(40) * (columns) stays far below the fp32 range
constexpr float kNegBig = -1.0e30f;     // "no element yet" stabiliser (finite: -inf - -inf would be NaN)

struct LSE {                            // running log-sum-exp state: z = sum of exp(x - m)
    float m, z;
    __device__ __forceinline__ void init() { m = kNegBig; z = 0.f; }
    __device__ __forceinline__ void add_exact(float x) {       // online update with the maximum as stabiliser (x may be -inf: no-op)
        if (x > m) { z = z * __expf(m - x) + 1.0f; m = x; }
        else z += __expf(x - m);
    }
    __device__ __forceinline__ void merge(float m2, float z2) {
        const float mn = fmaxf(m, m2);
        z = z * __expf(m - mn) + z2 * __expf(m2 - mn);
        m = mn;
    }
    // logsumexp_stable: m + log(max(sum, 1e-30))
    __device__ __forceinline__ float value() const { return m + __logf(fmaxf(z, 1e-30f)); }
};

template <int CTRL> __device__ __forceinline__ int dpp_movi(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
// reductions over aligned groups of GL lanes, result in every lane of the group: 4 DPP steps cover 16 lanes, ds_bpermute the rest
template <int GL> __device__ __forceinline__ float group_sum_f(float x) {
    x += dpp_mov<0xB1>(x); x += dpp_mov<0x4E>(x); x += dpp_mov<0x141>(x); x += dpp_mov<0x140>(x);
    if (GL >= 32) x += __shfl_xor(x, 16, 64);
    if (GL >= 64) x += __shfl_xor(x, 32, 64);
    return x;
}
template <int GL> __device__ __forceinline__ float group_max_f(float x) {
    x = fmaxf(x, dpp_mov<0xB1>(x)); x = fmaxf(x, dpp_mov<0x4E>(x)); x = fmaxf(x, dpp_mov<0x141>(x)); x = fmaxf(x, dpp_mov<0x140>(x));
    if (GL >= 32) x = fmaxf(x, __shfl_xor(x, 16, 64));
    if (GL >= 64) x = fmaxf(x, __shfl_xor(x, 32, 64));
    return x;
}
template <int GL> __device__ __forceinline__ int group_min_i(int x) {
    x = min(x, dpp_movi<0xB1>(x)); x = min(x, dpp_movi<0x4E>(x)); x = min(x, dpp_movi<0x141>(x)); x = min(x, dpp_movi<0x140>(x));
    if (GL >= 32) x = min(x, __shfl_xor(x, 16, 64));
    if (GL >= 64) x = min(x, __shfl_xor(x, 32, 64));
    return x;
}

// 16-byte pieces in front of row i of the LDS-resident triangle (row r holds its columns 0 .. r, rounded up to whole pieces)
__device__ __forceinline__ int tri_pieces(int i, int ppe) {        // ppe: elements per piece
    const int q = i / ppe, rem = i - q * ppe;
    return i + ppe * ((q * (q - 1)) >> 1) + q * rem;               // sum_{r < i} (floor(r / ppe) + 1)
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
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = w + 1;                                   // padded size
    const int ns = (n + 3) & ~3;                           // row stride of the LDS vectors (16-byte aligned rows)
    float* u = reinterpret_cast<float*>(smem);             // [ns]
    float* v = u + ns;                                     // [ns]
    float* pm = v + ns;                                    // [NWV][ns]  per-wave column partial stabiliser
    float* pz = pm + NWV * ns;                             // [NWV][ns]  per-wave column partial sum
    // [TRI] packed lower triangle behind the vectors and the three flag words, 16-byte aligned
    raw16_t* tri = reinterpret_cast<raw16_t*>(smem + (((size_t)(2 + 2 * NWV) * ns * sizeof(float) + 16 + 15) & ~(size_t)15));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = lane / GL, pl = lane % GL;             // row group inside the wave, position inside the group
    const TI* S = cv + (size_t)blockIdx.x * w * pitch;          // volume rows `pitch` elements apart (>= w, multiple of 8)
    const float log_row = -__logf(2.0f * w);               // log(1/(2w))   marginal of a regular row/column
    const float log_bin = __logf(0.5f);                    // log(w/(2w))   marginal of the dustbin
    const float log2w = __logf(2.0f * w);
    float* od = disp + (size_t)blockIdx.x * w;
    float* oc = conf + (size_t)blockIdx.x * w;
    float* oo = occ + (size_t)blockIdx.x * w;

    // row i (i == w: the dustbin row, S = 0, never masked): raw 16-byte pieces of the lane's columns j = c*CW + pl*8 + 0..7
    // SRC 0: global memory; 1: global memory + copy into the LDS triangle (pass 0 of a TRI block); 2: the LDS triangle
    auto fetch_row_from = [&](int i, raw16_t (&raw)[NCH][PPC], int src) __attribute__((always_inline)) {
        const TI* Si = S + (size_t)(i < w ? i : 0) * pitch;
        const int jend = i < w ? (use_pos ? i + 1 : w) : 0;
        raw16_t* trow = tri;
        if constexpr (TRI) trow = tri + tri_pieces(i < w ? i : 0, VEC);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < PPC; ++q) {
                const int j0 = c * CW + pl * 8 + q * VEC;
                if (j0 < jend) {                                            // w % 8 == 0: a piece starting inside [0, w) is whole
                    if (TRI && src == 2) raw[c][q] = trow[j0 / VEC];
                    else {
                        raw[c][q] = global_load16(Si + j0);
                        if (TRI && src == 1) trow[j0 / VEC] = raw[c][q];
                    }
                }
            }
    };
    auto fetch_row = [&](int i, raw16_t (&raw)[NCH][PPC]) __attribute__((always_inline)) { fetch_row_from(i, raw, TRI ? 2 : 0); };
    // nfast: leading chunks whose columns are valid for EVERY row of this wave in this step (regular rows, nothing masked): a plain
    // convert.  nchw: chunks that hold any unmasked column of the wave (later ones are never touched by any sweep, see chunks_needed).
    // The per-element compare / select (two instructions per element) is left to the one or two chunks the diagonal crosses -- measured
    // before this split: 126 of the ~410 VALU instructions of a 4-row step were the decode of 24 elements per lane, half of them masked.
    auto decode_row = [&](int i, const raw16_t (&raw)[NCH][PPC], float (&x)[NE], int nfast, int nchw) __attribute__((always_inline)) {
        const int jend = i < w ? (use_pos ? i + 1 : w) : (i == w ? w : 0);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c >= nchw) continue;                              // (wave-uniform)
            if (c < nfast) {
#pragma unroll
                for (int q = 0; q < PPC; ++q) {
                    const Vec16<TI> r = __builtin_bit_cast(Vec16<TI>, raw[c][q]);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) x[c * 8 + q * VEC + e] = to_f32(r.v[e]);
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < PPC; ++q) {
                const Vec16<TI> r = __builtin_bit_cast(Vec16<TI>, raw[c][q]);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int j = c * CW + pl * 8 + q * VEC + e;
                    x[c * 8 + q * VEC + e] = j < jend ? (i < w ? to_f32(r.v[e]) : 0.f) : -INFINITY;   // masked triangle / past the row
                }
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
    int* flags = reinterpret_cast<int*>(pz + NWV * ns);
    if (tid < 3) flags[tid] = 0;                           // (ordered before the first use by the barriers of pass 0)
    // the padding of v (entries n .. ns-1) is never written by the sweeps but IS read: the 16-byte read of v that holds the dustbin entry
    // v[w] also covers v[w+1 .. w+3], added to masked (-inf) columns -- whatever the previous kernel left in LDS there (+inf / NaN patterns)
    // would turn exp(-inf + garbage - m) into NaN and poison the row sum
    if (tid < ns - n) { v[n + tid] = 0.f; u[n + tid] = 0.f; }

    // ---- exact column sweep: v_j = log nu_j - LSE_i(S_ij + u_i) with lazy-maximum accumulators (pass 0, where u = 0, and the
    // fallback of the fast sweep below)
    auto exact_cols = [&](bool use_u) __attribute__((always_inline)) {        // (use_u false = pass 0: the one sweep that reads global memory in a TRI block)
        LSE col[NE], bin;
#pragma unroll
        for (int c = 0; c < NE; ++c) col[c].init();
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
            if (active) fetch_row_from(i, raw, use_u ? (TRI ? 2 : 0) : (TRI ? 1 : 0));
            float x[NE];
            decode_row(active ? i : w + 1, raw, x, chunks_fast(i0), nchw);
            const float ua = active ? (use_u ? u[i] : 0.f) : -INFINITY;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c >= nchw) continue;
                float t[8], dmax = -INFINITY;
#pragma unroll
                for (int e = 0; e < 8; ++e) { t[e] = x[c * 8 + e] + ua; dmax = fmaxf(dmax, t[e] - col[c * 8 + e].m); }
                if (__builtin_amdgcn_ballot_w64(dmax > kLazy)) {                  // rare: a stabiliser is too far below its new element
#pragma unroll
                    for (int e = 0; e < 8; ++e) col[c * 8 + e].add_exact(t[e]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) col[c * 8 + e].z += __expf(t[e] - col[c * 8 + e].m);
                }
            }
            bin.add_exact(ua);
        }
        // combine: across the row groups of a wave (lanes with equal pl), then across waves through LDS
#pragma unroll
        for (int off = GL; off < 64; off <<= 1) {
#pragma unroll
            for (int c = 0; c < NE; ++c) col[c].merge(__shfl_xor(col[c].m, off, 64), __shfl_xor(col[c].z, off, 64));
            bin.merge(__shfl_xor(bin.m, off, 64), __shfl_xor(bin.z, off, 64));
        }
        __syncthreads();                                   // readers of pm / pz / v of the previous pass are done
        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < NE; ++c) {
                const int j = (c / 8) * CW + pl * 8 + (c % 8);
                if (j < w) { pm[wv * ns + j] = col[c].m; pz[wv * ns + j] = col[c].z; }
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

    // ---- passes 1 .. ot_iter: row sweep (exact max-then-sum) fused with the next column sweep.  With e_ij = exp(S_ij + v_j - m_i)
    // from the row sweep, the column term is exp(S_ij + u_i - (c0 - v_j)) = e_ij * exp(u_i + m_i - c0): ONE multiply-add per element
    // against the per-row factor f_i -- the stabiliser c0 - v_j is the column's previous log-sum-exp up to a constant, and
    // f_i <= 1 for c0 = log(1/2) (u_i + m_i <= log mu_i), so nothing can overflow; a column whose sum underflows (it would need a
    // dynamic range of e^80 inside the volume) raises a flag and the pass is redone with exact_cols.
    const float c0 = log_bin;
    for (int pass = 1; pass <= ot_iter; ++pass) {
        const bool last = pass == ot_iter;
        float zc[NE], zbin = 0.f;
#pragma unroll
        for (int c = 0; c < NE; ++c) zc[c] = 0.f;
        const float vbin = v[w];
        int* flag = flags + pass % 3;
        if (tid == 0) flags[(pass + 1) % 3] = 0;
        raw16_t rcur[NCH][PPC], rnext[NCH][PPC];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int q = 0; q < PPC; ++q) { rcur[c][q] = (raw16_t){0.f, 0.f, 0.f, 0.f}; rnext[c][q] = rcur[c][q]; }
        fetch_row(r0, rcur);
        for (int i0 = 0; i0 <= w; i0 += RPB) {             // uniform trip count for every wave (groups past row w idle)
            const int i = i0 + r0;
            const bool active = i <= w;
            const int nchw = chunks_needed(i0);
            if (i + RPB <= w) fetch_row(i + RPB, rnext);          // in flight under this row's arithmetic
            float x[NE];
            decode_row(active ? i : w + 1, rcur, x, chunks_fast(i0), nchw);
#pragma unroll
            for (int c = 0; c < NCH; ++c)
#pragma unroll
                for (int q = 0; q < PPC; ++q) rcur[c][q] = rnext[c][q];
            // ---- row sweep: u_i = log mu_i - LSE_j(S_ij + v_j), dustbin column included (S = 0)
            float mloc = -INFINITY;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c >= nchw) continue;
                const int j0 = c * CW + pl * 8;
                const float4_t va = j0 < n ? *reinterpret_cast<const float4_t*>(v + j0) : float4_t{0.f, 0.f, 0.f, 0.f};
                const float4_t vb = j0 + 4 < n ? *reinterpret_cast<const float4_t*>(v + j0 + 4) : float4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[c * 8 + e] += e < 4 ? va[e] : vb[e - 4]; mloc = fmaxf(mloc, x[c * 8 + e]); }   // x <- S + v (masked: -inf)
            }
            const float m = fmaxf(group_max_f<GL>(mloc), vbin);
            float sloc = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (c >= nchw) continue;
#pragma unroll
                for (int e = 0; e < 8; ++e) { x[c * 8 + e] = __expf(x[c * 8 + e] - m); sloc += x[c * 8 + e]; }               // x <- e_ij
            }
            const float srow = group_sum_f<GL>(sloc);
            const float ebin = __expf(vbin - m);
            const float ui = (i == w ? log_bin : log_row) - (m + __logf(fmaxf(srow + ebin, 1e-30f)));
            if (!last) {
                if (pl == 0 && active) u[i] = ui;                                 // (only the exact fallback reads it)
                const float f = active ? __expf(ui + m - c0) : 0.f;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c >= nchw) continue;
#pragma unroll
                    for (int e = 0; e < 8; ++e) zc[c * 8 + e] = __builtin_fmaf(x[c * 8 + e], f, zc[c * 8 + e]);
                }
                zbin = __builtin_fmaf(ebin, f, zbin);
            } else if (i < w) {
                // ---- probabilities P_ij = e_ij * exp(m_i + u_i + log 2w), argmax (first max wins), window regression, row mass
                const float ci = ui + log2w;
                const float gsc = __expf(m + ci);
                const int jend = use_pos ? i + 1 : w;
                float best = -1.f;
                int bj = 0x7fffffff;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    if (c >= nchw) continue;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int j = c * CW + pl * 8 + e;
                        const float pr = x[c * 8 + e] * gsc;
                        if (j < jend && pr > best) { best = pr; bj = j; }        // strict: keeps the first maximum of this lane
                    }
                }
                const float bmax = group_max_f<GL>(best);
                bj = group_min_i<GL>(best == bmax ? bj : 0x7fffffff);
                const float mass = srow * gsc;
                // 5 taps around the argmax, evaluated by lanes 0..4 of the group (zero outside [0,w) and in the masked triangle)
                const TI* Si = TRI ? reinterpret_cast<const TI*>(tri + tri_pieces(i, VEC)) : S + (size_t)i * pitch;
                const int jj = bj + pl - 2;
                float pk = 0.f;
                if (pl < 5 && jj >= 0 && jj < jend) pk = __expf(to_f32(Si[jj]) + ci + v[jj]);
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
        }
        if (last) break;
        // ---- combine the partial column sums: across the row groups of a wave (lanes with equal pl), then across waves through LDS
#pragma unroll
        for (int off = GL; off < 64; off <<= 1) {
#pragma unroll
            for (int c = 0; c < NE; ++c) zc[c] += __shfl_xor(zc[c], off, 64);
            zbin += __shfl_xor(zbin, off, 64);
        }
        __syncthreads();                                   // every row sweep of this pass has read v
        if (grp == 0) {
#pragma unroll
            for (int c = 0; c < NE; ++c) {
                const int j = (c / 8) * CW + pl * 8 + (c % 8);
                if (j < w) pz[wv * ns + j] = zc[c];
            }
            if (pl == 0) pz[wv * ns + w] = zbin;
        }
        __syncthreads();
        for (int j = tid; j < n; j += NWV * 64) {
            float z = 0.f;
#pragma unroll
            for (int k = 0; k < NWV; ++k) z += pz[k * ns + j];
            if (!(z > 1e-30f) || !(z < 3.0e38f)) *flag = 1;                       // underflow (or NaN): this pass needs the exact sweep
            v[j] = (j == w ? log_bin : log_row) - ((c0 - v[j]) + __logf(z));
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
    auto kern = sinkhorn_regress_kernel<TI, NWV, GL, NCH, TRI>;
    size_t lds = (size_t)(2 + 2 * NWV) * ((w + 4) & ~3) * sizeof(float) + 16;
    if (TRI) {
        constexpr int VEC = 16 / sizeof(TI);
        const size_t pieces = (size_t)w + (size_t)VEC * ((size_t)(w / VEC) * (w / VEC - 1) / 2);       // tri_pieces(w): w % VEC == 0
        const size_t tri_bytes = ((lds + 15) & ~(size_t)15) + pieces * 16;
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

// lanes per row so that a row needs at most 3 chunks of 8 columns per lane: 16 lanes up to w = 384, 32 up to 768, 64 up to 1536
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

