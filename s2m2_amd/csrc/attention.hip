// K4 -- flash-style multi-head attention on MFMA: softmax(Q K^T * scale) V without materialising the score matrix.
//
// Replaces every F.scaled_dot_product_attention call and the explicit attention + positional-encoding einsums of the
// reference (attentions.py:42-50 SelfAttn, :83-91 CrossAttn; used by the 1-D epipolar blocks at 1/4, 1/8, 1/16 and the
// 2-D global blocks at 1/32 of stacked_MRT.py / unet.py).
//
// Layout: q, k, v, out are token rows (channels contiguous); element (batch b, token n, head hd, e) lives at
//   base + (b*N + n)*row_stride + hd*D + e, so the kernel reads the fused QKV projection in place and writes (tokens, C).
// Cross attention ("swap" = 1): keys/values of batch b come from batch (b + nb/2) % nb -- the symmetric left<->right
// attention of CrossAttn (shared weights, both directions in one launch).
//
//   block = NW waves = NW consecutive 32-query tiles of one (batch, head); K and V stream through LDS in tiles of 32 keys,
//           loaded by all waves (global -> registers before the MFMAs of the current tile, registers -> LDS after them).
//   S^T   = K_tile . Q^T  (v_mfma 32x32: a lane owns ONE query column and 16 of the 32 keys; the other 16 sit in lane^32),
//           so the online-softmax row maximum / sum are in-lane reductions plus one cross-half exchange.
//   O^T  += V_tile^T . P^T: the probabilities go straight from the S accumulators into the B operand (the k-slot -> key
//           map of a register octet is {16s+4hi+0..3, 16s+8+4hi+0..3}); V is staged TRANSPOSED in LDS (Vt[d][key]) so the
//           matching A operand is two 4-element reads per lane.  No permutes, no P round trip through LDS.
//   PE    = (template) contextual relative positional encoding of SelfAttn(use_pe): pe_sum[q,:] = sum_k P[q,k] pe[q,k,:]
//           with pe[q,k] = 0.5*[px[xq-xk+w-1], py[yq-yk+h-1]] from the two separable sinc tables (utils.py:32-60) kept in LDS
//           -- the (N,N,32) tensor of the reference (95 MB at 1216x1024, 1.5 GB for XL) is never built.  Because the encoding is
//           separable, the sum over the N keys factors through the MARGINALS of the attention row over key columns and key rows:
//             pe_x[q,:] = sum_x' ( sum_{k: x_k = x'} P[q,k] ) px[xq - x' + w - 1, :],   pe_y likewise over y'.
//           The marginals are two more MFMA products per key tile -- bins^T += OneHot^T . P^T with a one-hot (key -> x bin / y bin)
//           A operand built from integer compares, the SAME P^T fragments as the PV product as B operand, fp32 accumulators next
//           to O -- instead of 32 multiply-adds and 8 table reads per key on the VALU; the (w + h) x 16 contraction with the tables
//           runs once per query after the key loop.  (A first version binned with ds_add_f32: twice SLOWER than the VALU loop.)
// fp16: v_mfma_f32_32x32x16_f16 with fp32 softmax/accumulators;  fp32 (parity mode): exact v_mfma_f32_32x32x2_f32.
#include "common.h"
#include "plan.h"

#ifndef S2M2_ATTN_DBG
#define S2M2_ATTN_DBG 0          // ablation builds (timing only, wrong results): 1 no K/V refetch after stage 0, 2 no softmax math, 4 no MFMA, 8 no stage barriers
#endif

namespace s2m2 {

constexpr size_t kLdsBytes = 160 * 1024;

struct AttnArgs {
    const void* q; const void* k; const void* v; void* out;
    long long sq, sk, sv, so;           // row strides (elements)
    int nb, heads, Nq, Nk, D, swap;
    float scale;
    // positional encoding (PE variant only)
    const float* px; const float* py; void* pe_out; long long spe; int gw, gh;
    int dry;                            // host side only: plan the launch (tile variant, waves, LDS) and return without launching
    int nblk;                           // query blocks per (batch, head): the grid is 1-D, nblk * nb * heads blocks
    int xcd;                            // 1: XCD-aware block order (see attention_kernel)
};

template <int V> struct IntC { static constexpr int value = V; };       // compile-time mode argument of the key-loop lambda

template <typename T, int DP_, bool PE_, int MINW_ = 4, bool KSPLIT_ = false, int NXT_ = 2, int NYT_ = 1>
struct AttnCfg {
    static constexpr int NXT = NXT_, NYT = NYT_;         // PE: 32-wide bin tiles along x / y of the token grid (grid up to 32*NXT x 32*NYT)
    static constexpr int SM = (NXT_ + NYT_) * 32 + 1;    // floats per query of the bin scratch (odd: conflict-free rows)
    static constexpr bool KSPLIT = KSPLIT_;              // the 4 waves of a block share ONE 32-query tile and split its keys (few, long rows)
    static constexpr int MINW = MINW_;                   // fewest waves a block is launched with (sizes the staging registers)
    static constexpr int DP = DP_;                       // head dim rounded up to a multiple of 16
    static constexpr bool PE = PE_;
    static constexpr int VEC = 16 / sizeof(T);
    static constexpr int KSTEPS = DP / 16;
    static constexpr int ND = (DP + 31) / 32;            // 32-wide output d tiles
    // 32-key sub-tiles staged per block barrier (latency amortisation).  Key-split blocks (few, long rows: the 2-D attention at 1/32, 1216
    // keys x 16 (batch, head) pairs) are latency chains of stage barriers with 4 MFMAs per wave between them: fp16 stages of 256 keys
    // (two sub-tiles per wave and stage) halve the number of barriers -- measured (profiles/r03/attnbench*.txt) 21.6 -> 20.2 us for
    // (2,8,1216,32), 17.7 -> 15.0 for (1,8,1216,32); NOT for the PE variant (52 -> 65 us: its tables + bins leave less LDS per block)
#ifndef S2M2_ATTN_KSPLIT_KT
#define S2M2_ATTN_KSPLIT_KT 8            // (experiment builds: 4 = the round-2 stages of 128 keys)
#endif
    // (measured and dropped in round 4, profiles/r04/attnbench.txt: stages of 128 keys and two QK accumulators at d = 128: 88 us vs 84 - 87)
    static constexpr int KT = (KSPLIT_ && !PE_ && sizeof(T) == 2 && DP_ <= 64) ? S2M2_ATTN_KSPLIT_KT : (DP <= 64 ? 4 : (DP <= 128 ? 2 : 1));
    static constexpr int KVT = 32 * KT;                  // keys per stage
    static constexpr int KRS = DP + VEC;                 // K tile row stride (elements)
    static constexpr int VRS = KVT + 4;                  // Vt row stride (elements): keys of one stage + pad
    static constexpr int VROWS = ND * 32;
    static constexpr size_t K_BYTES = (size_t)KVT * KRS * sizeof(T);
    static constexpr size_t V_BYTES = (size_t)VROWS * VRS * sizeof(T);
    static constexpr int KP = DP / VEC;                  // 16-byte pieces per K/V row
    static constexpr int MAXW = (DP * sizeof(T) >= 384) ? 4 : 8;   // waves per block: 512 registers per lane for the wide heads
    // d = 384 in fp16 (the XL model's 1/4 level): 12 output tiles = 192 accumulator registers, and the 24 Q fragments (96 registers) no longer
    // fit next to the staging registers -- the compiler kept them in SCRATCH, re-read per 32-key sub-tile through the vector-memory path and,
    // because vmcnt retires in order, behind the K / V prefetch of the NEXT stage (its whole latency exposed in every stage: 100 TF/s).
    // Here the wave's 32 query rows live in LDS (same row stride as the K tile) and are read as fragments like the K operand.
#ifndef S2M2_ATTN_TWOPASS
#define S2M2_ATTN_TWOPASS 1            // 0: A/B build, the online softmax for every head dim (the form up to round 5)
#endif
#ifndef S2M2_ATTN_QLDS
#define S2M2_ATTN_QLDS 1
#endif
    // Wide heads (d >= 192: the 1/4 level of the M / L / XL models) keep their 6-12 output tiles in accumulation registers (AGPRs).  The
    // online softmax's `O *= alpha` is a VALU operation on them, and although it almost never runs, the compiler makes VGPRs the home of
    // the tiles across the key loop and copies ALL of them to AGPRs and back around every stage's MFMAs (2 x 96-192 v_accvgpr moves per 32
    // keys; plus scratch at d = 384).  TWOPASS removes the multiply instead of fighting the allocator: a first sweep over K takes the exact
    // row maximum (QK^T only: a third more MFMA work, K staged twice), the second sweep runs softmax and PV against that fixed maximum --
    // the tiles are touched by MFMAs alone.  Mathematically the same softmax; no running-maximum rounding at all.
    // Measured (profiles/r05/ab_attn_twopass.txt, fp16, same box, against the online form with the d = 384 queries already in LDS):
    //   d = 192 (512,1,304):  207 -> 156 us (256 registers instead of 336: two blocks per CU)
    //   d = 384 (1024,1,608): 3290 -> 2273 us (no scratch; 5760 us in round 4 with the Q fragments in scratch as well)
    //   d = 256 (512,1,304):  223 -> 256 us (312 registers either way, one block per CU: the extra QK^T sweep is not paid back) -> stays online
    static constexpr bool TWOPASS = S2M2_ATTN_TWOPASS && !PE_ && !KSPLIT_ && (DP_ == 192 || DP_ >= 384);
    static constexpr bool QLDS = S2M2_ATTN_QLDS && !PE_ && !KSPLIT_ && sizeof(T) == 2 && DP_ >= 384;
    static constexpr size_t Q_OFF = (K_BYTES + V_BYTES + 15) / 16 * 16;
    static constexpr size_t Q_WAVE_BYTES = (size_t)32 * KRS * sizeof(T);
    // (measured and dropped, profiles/r04/ab_minwaves.txt: a second launch bound of three waves per SIMD takes the key-split kernel from 208 to 121
    // registers without a spill -- four blocks per CU instead of two -- and changes nothing: 20.4 us, 8.857 vs 8.858 ms per pair)
    // KSPLIT merge scratch (reuses the K/V staging area after the key loop): running max / sum + the four partial O tiles
    static constexpr size_t MERGE_BYTES = KSPLIT ? (size_t)(256 + 4 * 32 * (ND * 32 + 1)) * sizeof(float) : 0;
    // PE area (tables + marginal bins) sits behind BOTH, so that it survives the merge
    static constexpr size_t PE_OFF = ((K_BYTES + V_BYTES > MERGE_BYTES ? K_BYTES + V_BYTES : MERGE_BYTES) + 15) / 16 * 16;
};

template <typename T> struct Quad;                                     // 4 consecutive elements
template <> struct alignas(8) Quad<half_t> { half_t v[4]; };
template <> struct alignas(16) Quad<float> { float v[4]; };

template <typename T> __device__ __forceinline__ void make_pfrag(Frag<T>& f, const float* p);
template <> __device__ __forceinline__ void make_pfrag<half_t>(Frag<half_t>& f, const float* p) {
    half8_t h = {(half_t)p[0], (half_t)p[1], (half_t)p[2], (half_t)p[3], (half_t)p[4], (half_t)p[5], (half_t)p[6], (half_t)p[7]};
    f.v = h;
}
template <> __device__ __forceinline__ void make_pfrag<float>(Frag<float>& f, const float* p) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f.v[e] = p[e];
}
template <typename T> __device__ __forceinline__ void load_vfrag(Frag<T>& f, const T* lo, const T* hi4);
template <> __device__ __forceinline__ void load_vfrag<half_t>(Frag<half_t>& f, const half_t* a, const half_t* b) {
    const half4_t x = *reinterpret_cast<const half4_t*>(a);
    const half4_t y = *reinterpret_cast<const half4_t*>(b);
    half8_t h = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
    f.v = h;
}
template <> __device__ __forceinline__ void load_vfrag<float>(Frag<float>& f, const float* a, const float* b) {
    const float4_t x = *reinterpret_cast<const float4_t*>(a);
    const float4_t y = *reinterpret_cast<const float4_t*>(b);
    f.v[0] = x[0]; f.v[1] = x[1]; f.v[2] = x[2]; f.v[3] = x[3];
    f.v[4] = y[0]; f.v[5] = y[1]; f.v[6] = y[2]; f.v[7] = y[3];
}

template <typename CFG, typename T>
__global__ __launch_bounds__(CFG::MAXW * 64) void attention_kernel(AttnArgs a) {
    constexpr int VEC = CFG::VEC, KRS = CFG::KRS, VRS = CFG::VRS, ND = CFG::ND, KP = CFG::KP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* Ks = reinterpret_cast<T*>(smem);                                   // [KVT][KRS]
    T* Vt = reinterpret_cast<T*>(smem + CFG::K_BYTES);                    // [VROWS][VRS]
    float* pxs = reinterpret_cast<float*>(smem + CFG::PE_OFF);            // PE tables, then the marginal bins (PE only)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthr = blockDim.x;
    const int NW = nthr >> 6;
    const int hi = lane >> 5, l31 = lane & 31;
    // Block order: the query blocks of one (batch, head) -- and the heads of one batch row, whose Q / K / V share cache lines -- re-read the SAME
    // K / V rows.  Hardware places block i on XCD i % 8 (speed assumption only), so in plain dispatch order the 2-3 query blocks of a row
    // land on different XCDs and every one of them pulls the row's K / V through the fabric into its own L2 (d = 128, N = 304: 240 MB instead
    // of 80).  xcd_remap hands each XCD a contiguous range of logical ids: blocks that share K / V run on one XCD at about the same time.
    const int lid = a.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
    const int bh = lid / a.nblk, qblk = lid - bh * a.nblk;
    const int b = bh / a.heads, hd = bh - b * a.heads;
    const int bkv = a.swap ? (b + a.nb / 2) % a.nb : b;
    const int q0 = CFG::KSPLIT ? qblk * 32 : (qblk * NW + wv) * 32;
    const bool wave_active = q0 < a.Nq;

    const T* qb = static_cast<const T*>(a.q) + (long long)b * a.Nq * a.sq + hd * a.D;
    const T* kb = static_cast<const T*>(a.k) + (long long)bkv * a.Nk * a.sk + hd * a.D;
    const T* vb = static_cast<const T*>(a.v) + (long long)bkv * a.Nk * a.sv + hd * a.D;

    float* pys = nullptr;
    float* pm = nullptr;
    constexpr int SM = CFG::SM;
    int Lx = 0;
    if constexpr (CFG::PE) {
        Lx = 2 * a.gw - 1;
        const int Ly = 2 * a.gh - 1;
        pys = pxs + Lx * 16;
        for (int i = tid; i < Lx * 16; i += nthr) pxs[i] = a.px[i];
        for (int i = tid; i < Ly * 16; i += nthr) pys[i] = a.py[i];
        // bin scratch of this wave's 32 queries (filled from the accumulators after the key loop): row (wv*32 + query), columns
        // [0, 32*NXT) = key x, [32*NXT, 32*(NXT+NYT)) = key y
        pm = pys + Ly * 16 + (size_t)wv * 32 * CFG::SM;
    }

    // ---- Q fragments of this wave (B operand: lane = query, 8 consecutive d per k16 step), zero beyond D
    Frag<T> qf[CFG::QLDS ? 1 : CFG::KSTEPS];
    const T* qs_lane = nullptr;
    if constexpr (CFG::QLDS) {
        T* Qs = reinterpret_cast<T*>(smem + CFG::Q_OFF + (size_t)wv * CFG::Q_WAVE_BYTES);      // wave-private: no block barrier involved
        qs_lane = Qs + (size_t)l31 * KRS + hi * 8;
        if (wave_active) {
#pragma unroll 4
            for (int t = lane; t < 32 * KP; t += 64) {
                const int r = t / KP, pc = t - r * KP;
                int qi = q0 + r;
                qi = qi < a.Nq ? qi : a.Nq - 1;
                Vec16<T> v;
                if (pc * VEC < a.D) v = *reinterpret_cast<const Vec16<T>*>(qb + (long long)qi * a.sq + pc * VEC);
                else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) v.v[e] = (half_t)0.f;
                }
                *reinterpret_cast<Vec16<T>*>(Qs + (size_t)r * KRS + pc * VEC) = v;
            }
        }
    } else {
        int qi = q0 + l31;
        qi = qi < a.Nq ? qi : a.Nq - 1;
        const T* qp = qb + (long long)qi * a.sq;
#pragma unroll
        for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
            const int d0 = kk * 16 + hi * 8;
            if (d0 < a.D) load_frag(qf[kk], qp + d0);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    if constexpr (sizeof(T) == 2) qf[kk].v[e] = (half_t)0.f; else qf[kk].v[e] = 0.f;
                }
            }
        }
    }

    // ---- staging assignment
    // K stage: KVT rows x KP pieces.   V stage: KVT/4 key groups (4 keys) x KP pieces -> transposed quads.
    constexpr int KVT = CFG::KVT;
    constexpr int K_TASKS = KVT * KP, V_TASKS = (KVT / 4) * KP;
    constexpr int MINT = 64 * CFG::MINW;                        // blocks have >= MINW waves (launch_attn)
    constexpr int K_IT_MAX = (K_TASKS + MINT - 1) / MINT, V_IT_MAX = (V_TASKS + MINT - 1) / MINT;
    // key-split blocks (few, long rows; fp16, no PE) are chains of short stages -- two sub-tiles of arithmetic per wave between the barriers --
    // whose K / V fetch (requested one stage ahead) is not covered by that arithmetic: they keep TWO stages in flight in registers
    constexpr bool DEEP = CFG::KSPLIT && !CFG::PE && sizeof(T) == 2 && !(S2M2_ATTN_DBG & 16);
    Vec16<T> rkA[K_IT_MAX], rkB[DEEP ? K_IT_MAX : 1];
    Vec16<T> rvA[V_IT_MAX][4], rvB[DEEP ? V_IT_MAX : 1][4];
    auto zero16 = []() { Vec16<T> z;
#pragma unroll
        for (int e = 0; e < VEC; ++e) { if constexpr (sizeof(T) == 2) z.v[e] = (half_t)0.f; else z.v[e] = 0.f; }
        return z; };
    auto fetch_into = [&](auto& rk, auto& rv, int kv0) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < K_IT_MAX; ++it) {
            const int t = tid + it * nthr;
            if (t < K_TASKS) {
                const int r = t / KP, pc = t - r * KP;
                const int kv = kv0 + r;
                rk[it] = (kv < a.Nk && pc * VEC < a.D) ? *reinterpret_cast<const Vec16<T>*>(kb + (long long)kv * a.sk + pc * VEC) : zero16();
            }
        }
#pragma unroll
        for (int it = 0; it < V_IT_MAX; ++it) {
            const int t = tid + it * nthr;
            if (t < V_TASKS) {
                const int kg = t / KP, pc = t - kg * KP;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int kv = kv0 + 4 * kg + j;
                    rv[it][j] = (kv < a.Nk && pc * VEC < a.D) ? *reinterpret_cast<const Vec16<T>*>(vb + (long long)kv * a.sv + pc * VEC) : zero16();
                }
            }
        }
    };
    auto stash_from = [&](auto& rk, auto& rv) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < K_IT_MAX; ++it) {
            const int t = tid + it * nthr;
            if (t < K_TASKS) {
                const int r = t / KP, pc = t - r * KP;
                *reinterpret_cast<Vec16<T>*>(Ks + (size_t)r * KRS + pc * VEC) = rk[it];
            }
        }
#pragma unroll
        for (int it = 0; it < V_IT_MAX; ++it) {
            const int t = tid + it * nthr;
            if (t < V_TASKS) {
                const int kg = t / KP, pc = t - kg * KP;
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    Quad<T> qd;
#pragma unroll
                    for (int j = 0; j < 4; ++j) qd.v[j] = rv[it][j].v[e];
                    *reinterpret_cast<Quad<T>*>(Vt + (size_t)(pc * VEC + e) * VRS + 4 * kg) = qd;
                }
            }
        }
    };

    auto fetch_k = [&](auto& rk, int kv0) __attribute__((always_inline)) {       // the K half of fetch_into / stash_from (TWOPASS, first sweep)
#pragma unroll
        for (int it = 0; it < K_IT_MAX; ++it) {
            const int t = tid + it * nthr;
            if (t < K_TASKS) {
                const int r = t / KP, pc = t - r * KP;
                const int kv = kv0 + r;
                rk[it] = (kv < a.Nk && pc * VEC < a.D) ? *reinterpret_cast<const Vec16<T>*>(kb + (long long)kv * a.sk + pc * VEC) : zero16();
            }
        }
    };
    auto stash_k = [&](auto& rk) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < K_IT_MAX; ++it) {
            const int t = tid + it * nthr;
            if (t < K_TASKS) {
                const int r = t / KP, pc = t - r * KP;
                *reinterpret_cast<Vec16<T>*>(Ks + (size_t)r * KRS + pc * VEC) = rk[it];
            }
        }
    };

    float16_t oacc[ND];
#pragma unroll
    for (int dt = 0; dt < ND; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[dt][r] = 0.f;
    if constexpr (CFG::TWOPASS) {
        // the tiles start life as MFMA results (0 . 0 + 0): with every value that reaches the key loop's phi defined in an accumulation
        // register, the compiler makes the phi an AGPR (SIFoldOperands::tryFoldPhiAGPR); zeros built by moves make it a VGPR phi with a
        // copy of every tile in each direction per stage
        Frag<T> zf;
#pragma unroll
        for (int e = 0; e < 8; ++e) { if constexpr (sizeof(T) == 2) zf.v[e] = (half_t)0.f; else zf.v[e] = 0.f; }
        if constexpr (sizeof(T) == 2) asm volatile("" : "+v"(zf.v));   // (opaque: a constant-folded product would be moves again)
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(zf.v[e]));
        }
#pragma unroll
        for (int dt = 0; dt < ND; ++dt) mma32(oacc[dt], zf, zf);
    }
    float16_t bacc[CFG::PE ? CFG::NXT + CFG::NYT : 1];        // PE: marginal bins^T [bin][query], x tiles then y tiles
    if constexpr (CFG::PE) {
#pragma unroll
        for (int t = 0; t < CFG::NXT + CFG::NYT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) bacc[t][r] = 0.f;
    }
    const float inv_gw = CFG::PE ? 1.0f / (float)a.gw : 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float scale2 = a.scale * 1.44269504088896340736f;
    // query grid coordinates (PE)
    int qi_pe = q0 + l31;
    qi_pe = qi_pe < a.Nq ? qi_pe : a.Nq - 1;
    const int yq = CFG::PE ? qi_pe / a.gw : 0, xq = CFG::PE ? qi_pe - yq * a.gw : 0;

    const int nstage = (a.Nk + KVT - 1) / KVT;
    // MODE 0: online softmax (running maximum, O rescaled when it moves); 1: row maximum only (first sweep of TWOPASS); 2: softmax + PV
    // against the final maximum left in m_run by the first sweep
    auto compute_stage = [&](int t, auto mode_c) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_c)::value;
        if (wave_active) {
#pragma unroll 1
            for (int sub = CFG::KSPLIT ? wv : 0; sub < CFG::KT; sub += CFG::KSPLIT ? 4 : 1) {
                const int kv0 = t * KVT + sub * 32;
                if (kv0 >= a.Nk) break;
                // ---- S^T = K . Q^T
                float16_t sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc[r] = 0.f;
                const T* kp = Ks + (size_t)(sub * 32 + l31) * KRS + hi * 8;
#pragma unroll
                for (int kk = 0; kk < CFG::KSTEPS; ++kk) {
                    Frag<T> kf;
                    load_frag(kf, kp + kk * 16);
                    if constexpr (CFG::QLDS) {
                        Frag<T> qk;
                        load_frag(qk, qs_lane + kk * 16);
                        mma32(sacc, kf, qk);
                    } else
                    if (S2M2_ATTN_DBG & 4) { sacc[kk & 15] += (float)kf.v[0] * (float)qf[kk].v[0]; } else
                    mma32(sacc, kf, qf[kk]);
                }
                // ---- online softmax (lane: one query, keys crow(r, hi) of this sub-tile)
                // scores in the log2 domain (scale * log2(e) folded into one multiply, v_exp_f32 = 2^x): m_run, m_new are log2-domain maxima
                // (the maximum is taken on the raw scores and scaled once -- scale2 > 0 --, and scale and shift are ONE fma per element:
                // 4 instead of 5 VALU operations per score)
                float p[16];
                float smax = -INFINITY;
                const bool full = kv0 + 32 <= a.Nk;                // (wave-uniform) a whole sub-tile of keys: no per-element bound check
                if (full) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) smax = fmaxf(smax, sacc[r]);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + acc_row(r, lane);
                        sacc[r] = kv < a.Nk ? sacc[r] : -INFINITY;
                        smax = fmaxf(smax, sacc[r]);
                    }
                }
                smax = fmaxf(smax, __shfl_xor(smax, 32, 64));
                if constexpr (MODE == 1) { m_run = fmaxf(m_run, smax * scale2); continue; }
                const float m_new = MODE == 2 ? m_run : fmaxf(m_run, smax * scale2);   // finite: every sub-tile has at least one valid key
                const float alpha = MODE == 2 ? 1.0f : __builtin_amdgcn_exp2f(m_run - m_new);        // 2^(-inf) = 0 on the first tile
                float lsum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    p[r] = (S2M2_ATTN_DBG & 2) ? sacc[r] : __builtin_amdgcn_exp2f(__builtin_fmaf(sacc[r], scale2, -m_new));   // (-inf * scale2 - m = -inf: 0)
                    lsum += p[r];
                }
                l_run = l_run * alpha + lsum;
                m_run = m_new;
                if constexpr (MODE == 0) {
                    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {             // the running maximum rarely moves after the first tiles
#pragma unroll
                        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
                            for (int r = 0; r < 16; ++r) oacc[dt][r] *= alpha;
                    }
                }
                // ---- O^T += Vt . P^T
                Frag<T> pf[2];
                make_pfrag<T>(pf[0], p);
                make_pfrag<T>(pf[1], p + 8);
#pragma unroll
                for (int dt = 0; dt < ND; ++dt) {
                    const T* vp = Vt + (size_t)(dt * 32 + l31) * VRS + sub * 32 + 4 * hi;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        Frag<T> vf;
                        load_vfrag<T>(vf, vp + 16 * s, vp + 16 * s + 8);
                        if (S2M2_ATTN_DBG & 4) { oacc[dt][s] += (float)vf.v[0] * (float)pf[s].v[0]; } else
                        mma32(oacc[dt], vf, pf[s]);
                    }
                }
                if constexpr (CFG::PE) {
                    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f)) {
#pragma unroll
                        for (int t = 0; t < CFG::NXT + CFG::NYT; ++t)
#pragma unroll
                            for (int r = 0; r < 16; ++r) bacc[t][r] *= alpha;
                    }
                    // bins^T[bin][q] += OneHot^T[bin][key] . P^T[key][q]: the B operand is pf[s] (fp16 mode: the probabilities rounded to
                    // fp16, as the reference's autocast einsum sees them); the A operand of lane (bin row l31 of tile t, half hi) holds,
                    // for its 8 k slots = keys 16s + 4hi + {0..3} and 16s + 8 + 4hi + {0..3}, 1 where the key falls into that bin
#pragma unroll
                    for (int sk = 0; sk < 2; ++sk) {
                        int xk[8], yk[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int kv = kv0 + 16 * sk + 4 * hi + (e & 3) + 8 * (e >> 2);
                            int y = (int)(((float)kv + 0.5f) * inv_gw);          // kv / gw: exact for kv < 2^20
                            yk[e] = y;
                            xk[e] = kv - y * a.gw;
                        }
#pragma unroll
                        for (int t = 0; t < CFG::NXT + CFG::NYT; ++t) {
                            const int bin = (t < CFG::NXT ? t : t - CFG::NXT) * 32 + l31;
                            Frag<T> oh;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const bool hit = (t < CFG::NXT ? xk[e] : yk[e]) == bin;
                                if constexpr (sizeof(T) == 2) oh.v[e] = hit ? (half_t)1.0f : (half_t)0.0f; else oh.v[e] = hit ? 1.0f : 0.0f;
                            }
                            mma32(bacc[t], oh, pf[sk]);
                        }
                    }
                }
            }
        }
    };
    if constexpr (DEEP) {
        // two stages in flight: stage t is stashed from one register set while stage t + 1 is still landing in the other; the set just stashed
        // is re-requested with stage t + 2 right away
        fetch_into(rkA, rvA, 0);
        if (nstage > 1) fetch_into(rkB, rvB, KVT);
        for (int t = 0; t < nstage; t += 2) {
            __syncthreads();                                      // previous stage fully consumed
            stash_from(rkA, rvA);
            __syncthreads();
            if (t + 2 < nstage) fetch_into(rkA, rvA, (t + 2) * KVT);
            compute_stage(t, IntC<0>{});
            if (t + 1 < nstage) {
                __syncthreads();
                stash_from(rkB, rvB);
                __syncthreads();
                if (t + 3 < nstage) fetch_into(rkB, rvB, (t + 3) * KVT);
                compute_stage(t + 1, IntC<0>{});
            }
        }
    } else {
        if constexpr (CFG::TWOPASS) {                                 // first sweep: K only, exact row maxima into m_run
            fetch_k(rkA, 0);
            for (int t = 0; t < nstage; ++t) {
                __syncthreads();
                stash_k(rkA);
                __syncthreads();
                if (t + 1 < nstage) fetch_k(rkA, (t + 1) * KVT);
                compute_stage(t, IntC<1>{});
            }
        }
        fetch_into(rkA, rvA, 0);
        for (int t = 0; t < nstage; ++t) {
            if (!(S2M2_ATTN_DBG & 8) || t == 0) __syncthreads();      // previous stage fully consumed (also covers the PE table fill)
            if (!(S2M2_ATTN_DBG & 1) || t == 0) stash_from(rkA, rvA);
            if (!(S2M2_ATTN_DBG & 8) || t == 0) __syncthreads();
            if (t + 1 < nstage && !(S2M2_ATTN_DBG & 1)) fetch_into(rkA, rvA, (t + 1) * KVT);   // in flight under this stage's MFMAs
            compute_stage(t, IntC<CFG::TWOPASS ? 2 : 0>{});
        }
    }

    if constexpr (CFG::KSPLIT) {
        // ---- the four waves hold partial (max, sum, O, pe) over disjoint key subsets of the SAME 32 queries: merge through LDS
        static_assert(CFG::KT % 4 == 0, "key split: every wave takes every fourth 32-key sub-tile of a stage");
        __syncthreads();                                          // K/V staging is dead: reuse the space (the PE area lies behind it)
        float* mm = reinterpret_cast<float*>(smem);               // [4][32] running max
        float* ml = mm + 128;                                     // [4][32] running sum
        float* mo = ml + 128;                                     // [4][32][DPO] partial O (DPO = ND*32 + 1: odd stride, conflict-free)
        constexpr int DPO = ND * 32 + 1;
        const float lw = l_run + __shfl_xor(l_run, 32, 64);
        if (hi == 0) { mm[wv * 32 + l31] = m_run; ml[wv * 32 + l31] = lw; }
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mo[(wv * 32 + l31) * DPO + dt * 32 + acc_row(r, lane)] = oacc[dt][r];
        if constexpr (CFG::PE) {
#pragma unroll
            for (int t = 0; t < CFG::NXT + CFG::NYT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) pm[(size_t)l31 * SM + t * 32 + acc_row(r, lane)] = bacc[t][r];
        }
        __syncthreads();
        // thread t: query t & 31, channel group t >> 5 (8 groups)
        const int q = tid & 31, grp = tid >> 5;
        const int qi = q0 + q;
        float mx = fmaxf(fmaxf(mm[q], mm[32 + q]), fmaxf(mm[64 + q], mm[96 + q]));
        float f[4], lsum = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) { f[w4] = __builtin_amdgcn_exp2f(mm[w4 * 32 + q] - mx); lsum += f[w4] * ml[w4 * 32 + q]; }   // log2 domain; 2^(-inf) = 0: idle wave
        const float inv = 1.0f / lsum;
        if (qi < a.Nq) {
            T* op = static_cast<T*>(a.out) + ((long long)b * a.Nq + qi) * a.so + hd * a.D;
            for (int d0 = grp * 4; d0 < a.D; d0 += 32) {
                Quad<T> qd;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = 0.f;
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) v += f[w4] * mo[(w4 * 32 + q) * DPO + d0 + e];
                    qd.v[e] = from_f32<T>(v * inv);
                }
                *reinterpret_cast<Quad<T>*>(op + d0) = qd;
            }
            if constexpr (CFG::PE) {
                // channels 4*grp .. +3: groups 0-3 = x part, 4-7 = y part.  Merged marginal of the query = sum over the four waves'
                // bins with the softmax merge factors, contracted with the table row selected by the relative offset.
                const float* base = pys + (2 * a.gh - 1) * 16;                // bins of wave 0
                const int yq2 = qi / a.gw, xq2 = qi - yq2 * a.gw;
                const bool ypart = grp >= 4;
                const int n = ypart ? a.gh : a.gw, off = ypart ? CFG::NXT * 32 : 0;
                const float* tab = ypart ? pys + (yq2 + a.gh - 1) * 16 + (grp - 4) * 4 : pxs + (xq2 + a.gw - 1) * 16 + grp * 4;
                float4_t acc4 = {0.f, 0.f, 0.f, 0.f};
                for (int j = 0; j < n; ++j) {
                    float mj = 0.f;
#pragma unroll
                    for (int w4 = 0; w4 < 4; ++w4) mj += f[w4] * base[(size_t)(w4 * 32 + q) * SM + off + j];
                    const float4_t tj = *reinterpret_cast<const float4_t*>(tab - j * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc4[e] = __builtin_fmaf(mj, tj[e], acc4[e]);
                }
                T* pp = static_cast<T*>(a.pe_out) + ((long long)b * a.Nq + qi) * a.spe + hd * 32 + grp * 4;
                Quad<T> qd;
#pragma unroll
                for (int e = 0; e < 4; ++e) qd.v[e] = from_f32<T>(0.5f * acc4[e] * inv);
                *reinterpret_cast<Quad<T>*>(pp) = qd;
            }
        }
        return;
    }
    if (!wave_active) return;
    // ---- normalise and store: lane owns query q0 + l31, d = 32*dt + 8*g + 4*hi + 0..3
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    const int qi = q0 + l31;
    if constexpr (CFG::PE) {                                      // bins of both halves -> this wave's scratch rows
#pragma unroll
        for (int t = 0; t < CFG::NXT + CFG::NYT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) pm[(size_t)l31 * SM + t * 32 + acc_row(r, lane)] = bacc[t][r];
        __builtin_amdgcn_wave_barrier();
    }
    if (qi < a.Nq) {
        T* op = static_cast<T*>(a.out) + ((long long)b * a.Nq + qi) * a.so + hd * a.D;
#pragma unroll
        for (int dt = 0; dt < ND; ++dt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = dt * 32 + 8 * g + 4 * hi;
                if (d < a.D) {
                    Quad<T> qd;
#pragma unroll
                    for (int e = 0; e < 4; ++e) qd.v[e] = from_f32<T>(oacc[dt][4 * g + e] * inv);
                    *reinterpret_cast<Quad<T>*>(op + d) = qd;
                }
            }
        if constexpr (CFG::PE) {
            // pe_sum (b, q, head, 32) in T; the 0.5 of get_pe is applied here; half hi = 0 contracts the x bins with the px rows
            // (channels 0..15), half hi = 1 the y bins with the py rows (channels 16..31)
            const float* row = pm + (size_t)l31 * SM + (hi ? CFG::NXT * 32 : 0);
            const int n = hi ? a.gh : a.gw;
            const float* tab = hi ? pys + (yq + a.gh - 1) * 16 : pxs + (xq + a.gw - 1) * 16;
            float4_t acc4[4];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) acc4[c4] = float4_t{0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < n; ++j) {
                const float mj = row[j];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    const float4_t tj = *reinterpret_cast<const float4_t*>(tab - j * 16 + c4 * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc4[c4][e] = __builtin_fmaf(mj, tj[e], acc4[c4][e]);
                }
            }
            T* pp = static_cast<T*>(a.pe_out) + ((long long)b * a.Nq + qi) * a.spe + hd * 32 + 16 * hi;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                Quad<T> qd;
#pragma unroll
                for (int e = 0; e < 4; ++e) qd.v[e] = from_f32<T>(0.5f * acc4[c4][e] * inv);
                *reinterpret_cast<Quad<T>*>(pp + c4 * 4) = qd;
            }
        }
    }
}

template <typename T, int DP, bool PE, int MINW, bool KSPLIT, int NXT, int NYT>
static int launch_attn_t(const AttnArgs& a, int nw, hipStream_t st) {
    using CFG = AttnCfg<T, DP, PE, MINW, KSPLIT, NXT, NYT>;
    auto kern = attention_kernel<CFG, T>;
    size_t lds = CFG::K_BYTES + CFG::V_BYTES;
    if (KSPLIT) lds = lds > CFG::MERGE_BYTES ? lds : CFG::MERGE_BYTES;
    if (CFG::QLDS) lds = CFG::Q_OFF + (size_t)nw * CFG::Q_WAVE_BYTES;
    if (PE) {                                                     // tables + (w + h | 1) marginal bins per query of every wave
        auto need = [&](int n) { return CFG::PE_OFF + ((size_t)(2 * a.gw - 1 + 2 * a.gh - 1) * 16 + (size_t)n * 32 * CFG::SM) * sizeof(float); };
        // large token grids: fewer waves per block until tables + bins fit next to the K/V stages (the block count follows nw below;
        // the key-split mode always runs its four waves)
        if (!KSPLIT) while (need(nw) > kLdsBytes && nw > MINW) --nw;
        lds = need(nw);
    }
    if (lds > kLdsBytes) return set_error("attention: %zu bytes of LDS needed (head dim %d, %d x %d token grid)", lds, a.D, a.gw, a.gh);
    if (a.dry) return 0;
    static size_t lds_granted[kMaxDevices] = {};                     // per instantiation
    if (reserve_lds(reinterpret_cast<const void*>(kern), lds, lds_granted, "attention")) return 1;
    const int ntq = (a.Nq + 31) / 32;
    const int nblk = KSPLIT ? ntq : (ntq + nw - 1) / nw;
    if ((long long)nblk * a.nb * a.heads >= (1LL << 31)) return set_error("attention: %lld blocks do not fit a grid dimension", (long long)nblk * a.nb * a.heads);
    AttnArgs b = a;
    b.nblk = nblk;
    static const bool xcd_off = [] { const char* e = getenv("S2M2_ATTN_XCD"); return e && e[0] == '0'; }();   // A/B switch (profiles/r05/ab_attn_xcd.txt)
    b.xcd = xcd_off ? 0 : 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nblk * a.nb * a.heads)), dim3(nw * 64), lds, st, b);
    return check_launch("attention");
}

// PE variant: bin tiles sized for the token grid -- (2, 1) covers grids up to 64 x 32 (the 1/32 grid of 2048 x 1024 images), (3, 2) up to
// 96 x 64 (3072 x 2048: BASELINE's XL 2432 x 2048 pair is 76 x 64), (3, 3) up to 96 x 96 (3072 x 3072); without PE the two parameters are
// inert.  engine.check_limits asks s2m2_attention_supported before the first launch of a forward.
template <typename T, int DP, bool PE, int MINW, bool KSPLIT = false>
static int launch_attn_w(const AttnArgs& a, int nw, hipStream_t st) {
    if constexpr (PE) {
        if (a.gw <= 64 && a.gh <= 32) return launch_attn_t<T, DP, PE, MINW, KSPLIT, 2, 1>(a, nw, st);
        if (a.gw <= 96 && a.gh <= 64) return launch_attn_t<T, DP, PE, MINW, KSPLIT, 3, 2>(a, nw, st);
        if (a.gw <= 96 && a.gh <= 96) return launch_attn_t<T, DP, PE, MINW, KSPLIT, 3, 3>(a, nw, st);
        return set_error("attention: positional-encoding grid %d x %d exceeds 96 x 96 cells", a.gw, a.gh);
    } else {
        return launch_attn_t<T, DP, PE, MINW, KSPLIT, 1, 1>(a, nw, st);
    }
}

// LDS the PE variant needs with nw waves per block (tables + marginal bins behind the K / V stages), for the bin tiles launch_attn_w picks
template <typename T, int DP, int MINW, bool KSPLIT>
static size_t pe_lds_need(const AttnArgs& a, int nw) {
    const int sm = (a.gw <= 64 && a.gh <= 32) ? AttnCfg<T, DP, true, MINW, KSPLIT, 2, 1>::SM
                   : (a.gh <= 64 ? AttnCfg<T, DP, true, MINW, KSPLIT, 3, 2>::SM : AttnCfg<T, DP, true, MINW, KSPLIT, 3, 3>::SM);
    return AttnCfg<T, DP, true, MINW, KSPLIT, 2, 1>::PE_OFF + ((size_t)(2 * a.gw - 1 + 2 * a.gh - 1) * 16 + (size_t)nw * 32 * sm) * sizeof(float);
}

template <typename T, int DP, bool PE>
static int launch_attn(const AttnArgs& a, hipStream_t st) {
    constexpr int MAXW = AttnCfg<T, DP, PE>::MAXW;
    const int ntq = (a.Nq + 31) / 32;
    const int bh = a.nb * a.heads;
    // register-heavy head dims (d >= 96: ~190-250 VGPRs = 8 waves per CU): blocks of at most 4 waves so that two are co-resident
    int maxw = (DP >= 96 && MAXW > 4) ? 4 : MAXW;
    static const int maxw_env = [] { const char* e = getenv("S2M2_ATTN_MAXW"); return e ? atoi(e) : 0; }();   // experiment switch
    if (maxw_env > 0 && maxw_env <= MAXW) maxw = maxw_env;
    int nblk = (ntq + maxw - 1) / maxw;
    int nw = (ntq + nblk - 1) / nblk;                              // <= maxw waves per block, minimal idle tail
    if constexpr (DP <= 64) {
        // few, long rows (the 2-D global blocks at 1/32: 8-16 (batch, head) pairs x 1216 tokens): a wave per query tile would leave
        // three quarters of the chip idle and walk all keys serially -> four waves per query tile, each taking every fourth 32-key
        // sub-tile, partial softmax states merged through LDS
        if ((long long)ntq * bh < 2048 && a.Nk >= 256) {
            if constexpr (PE) {
                // tall fp32 token grids (d = 64 on 76 x 64: 169 KB): tables + bins of the four key-split waves do not fit next to the
                // 128-key stages -> two-wave blocks without the key split
                if (pe_lds_need<T, DP, 4, true>(a, 4) > kLdsBytes) return launch_attn_w<T, DP, PE, 2>(a, 2, st);
            }
            return launch_attn_w<T, DP, PE, 4, true>(a, 4, st);
        }
        // few (batch, head) pairs (the 2-D global blocks at 1/32): smaller blocks until the grid covers the chip; every block
        // re-stages K/V from L2, which is cheap next to an idle GPU
        while (nw > 2 && (long long)((ntq + nw - 1) / nw) * bh < 512) nw = (nw + 1) / 2;
        if (nw < 4) return launch_attn_w<T, DP, PE, 2>(a, nw, st);
    }
    return launch_attn_w<T, DP, PE, 4>(a, nw < 4 ? 4 : nw, st);
}

template <typename T, bool PE>
static int dispatch_attn(const AttnArgs& a, hipStream_t st) {
    const int dp = (a.D + 15) / 16 * 16;
    switch (dp) {
        case 16: return launch_attn<T, 16, PE>(a, st);
        case 32: return launch_attn<T, 32, PE>(a, st);
        case 48: return launch_attn<T, 48, PE>(a, st);
        case 64: return launch_attn<T, 64, PE>(a, st);
        case 96: return launch_attn<T, 96, PE>(a, st);
        case 128: return launch_attn<T, 128, PE>(a, st);
        default: break;
    }
    if (!PE) {
        switch (dp) {
            case 192: return launch_attn<T, 192, false>(a, st);
            case 256: return launch_attn<T, 256, false>(a, st);
            case 384: return launch_attn<T, 384, false>(a, st);
            default: break;
        }
    }
    return set_error("attention: unsupported head dim %d", a.D);
}

}  // namespace s2m2

static int attention_entry(const void* q, const void* k, const void* v, void* out, long long q_stride, long long k_stride,
                           long long v_stride, long long out_stride, int nb, int heads, int Nq, int Nk, int D, float scale,
                           int swap_halves, const float* pe_x, const float* pe_y, void* pe_out, long long pe_stride,
                           int grid_w, int grid_h, int dtype, void* stream, int dry) {
    using namespace s2m2;
    S2M2_REQUIRE(dry || (q && k && v && out), "attention: null pointer");
    S2M2_REQUIRE(nb > 0 && heads > 0 && Nq > 0 && Nk > 0 && D > 0 && D % 8 == 0, "attention: bad shape nb=%d heads=%d Nq=%d Nk=%d D=%d", nb, heads, Nq, Nk, D);
    S2M2_REQUIRE(q_stride % 8 == 0 && k_stride % 8 == 0 && v_stride % 8 == 0 && out_stride % 4 == 0, "attention: strides must be multiples of 8");
    S2M2_REQUIRE(!swap_halves || (nb % 2 == 0 && Nq == Nk), "attention: swap_halves needs an even batch and Nq == Nk");
    const bool pe = pe_x != nullptr;
    if (pe) {
        S2M2_REQUIRE(pe_y && pe_out && grid_w > 0 && grid_h > 0 && grid_w * grid_h == Nq && Nq == Nk && pe_stride % 4 == 0,
                     "attention: PE variant needs py, pe_out and a grid with w*h == Nq == Nk");
    }
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.out = out; a.sq = q_stride; a.sk = k_stride; a.sv = v_stride; a.so = out_stride;
    a.nb = nb; a.heads = heads; a.Nq = Nq; a.Nk = Nk; a.D = D; a.swap = swap_halves; a.scale = scale;
    a.px = pe_x; a.py = pe_y; a.pe_out = pe_out; a.spe = pe_stride; a.gw = grid_w; a.gh = grid_h; a.dry = dry;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == S2M2_F16) return pe ? dispatch_attn<half_t, true>(a, st) : dispatch_attn<half_t, false>(a, st);
    if (dtype == S2M2_F32) return pe ? dispatch_attn<float, true>(a, st) : dispatch_attn<float, false>(a, st);
    return set_error("attention: unsupported dtype %d", dtype);
}

static int attention_impl(const void* q, const void* k, const void* v, void* out, long long q_stride, long long k_stride,
                              long long v_stride, long long out_stride, int nb, int heads, int Nq, int Nk, int D, float scale,
                              int swap_halves, const float* pe_x, const float* pe_y, void* pe_out, long long pe_stride,
                              int grid_w, int grid_h, int dtype, void* stream) {
    return attention_entry(q, k, v, out, q_stride, k_stride, v_stride, out_stride, nb, heads, Nq, Nk, D, scale, swap_halves, pe_x, pe_y,
                           pe_out, pe_stride, grid_w, grid_h, dtype, stream, 0);
}
extern "C" int s2m2_attention(const void* q, const void* k, const void* v, void* out, long long q_stride, long long k_stride,
                              long long v_stride, long long out_stride, int nb, int heads, int Nq, int Nk, int D, float scale,
                              int swap_halves, const float* pe_x, const float* pe_y, void* pe_out, long long pe_stride,
                              int grid_w, int grid_h, int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_attention", &attention_impl, stream, q, k, v, out, q_stride, k_stride, v_stride, out_stride, nb, heads, Nq, Nk, D, scale, swap_halves, pe_x, pe_y, pe_out, pe_stride, grid_w, grid_h, dtype);
}


extern "C" int s2m2_attention_supported(int nb, int heads, int N, int D, int grid_w, int grid_h, int dtype) {
    // the same planning code as the launch (head-dim instantiation, PE bin tiles, waves per block, LDS budget) without the launch;
    // grid_w = grid_h = 0: no positional encoding.  1 = supported, 0 = not (s2m2_last_error says why)
    static const float dummy = 0.f;
    const bool pe = grid_w > 0 || grid_h > 0;
    const int c = heads * ((D + 7) / 8 * 8);
    return attention_entry(nullptr, nullptr, nullptr, nullptr, 3 * c, 3 * c, 3 * c, c, nb, heads, N, N, D, 1.0f, 0, pe ? &dummy : nullptr,
                           pe ? &dummy : nullptr, pe ? const_cast<float*>(&dummy) : nullptr, heads * 32, grid_w, grid_h, dtype, nullptr, 1) == 0;
}
