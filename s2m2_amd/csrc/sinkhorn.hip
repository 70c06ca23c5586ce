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
// loads (a lane owns 8 fp16 / 4 fp32 consecutive columns): while a row is in registers the wave computes its u_i (exact
// max-then-sum log-sum-exp, DPP reductions) AND immediately feeds S_ij + u_i into per-lane online column accumulators for the next
// v -- the row sweep of iteration k and the column sweep of iteration k+1 are one pass; the last pass goes straight on to the
// probabilities, argmax (first maximum), row mass and window regression.  u, v and the per-wave column partials live in LDS.
#include "common.h"

namespace s2m2 {

struct LSE {                         // running log-sum-exp state: sum of exp(x - m)
    float m, z;
    __device__ __forceinline__ void init() { m = -INFINITY; z = 0.f; }
    __device__ __forceinline__ void add(float x) {             // branch-free online update (x may be -inf: no-op)
        const float mn = fmaxf(m, x);
        const float e = __expf(-fabsf(x - m));                  // exp(-inf) = 0 on the first element; NaN-free: (-inf) - (-inf) guarded
        const bool up = x > m;
        z = (x == -INFINITY) ? z : (up ? z * e + 1.0f : z + e);
        m = mn;
    }
    __device__ __forceinline__ void merge(float m2, float z2) {
        const float mn = fmaxf(m, m2);
        if (mn == -INFINITY) return;
        z = z * __expf(m - mn) + z2 * __expf(m2 - mn);
        m = mn;
    }
    // logsumexp_stable: m + log(max(sum, 1e-30))
    __device__ __forceinline__ float value() const { return m + __logf(fmaxf(z, 1e-30f)); }
};

// full-wave reductions on the VALU data path: 4 DPP steps inside each row of 16 lanes, 4 v_readlane across the rows
template <int CTRL> __device__ __forceinline__ int dpp_movi(int x) { return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ float rl(float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); }
__device__ __forceinline__ float wave_sum_dpp(float x) {
    x += dpp_mov<0xB1>(x); x += dpp_mov<0x4E>(x); x += dpp_mov<0x141>(x); x += dpp_mov<0x140>(x);
    return (rl(x, 0) + rl(x, 16)) + (rl(x, 32) + rl(x, 48));
}
__device__ __forceinline__ float wave_max_dpp(float x) {
    x = fmaxf(x, dpp_mov<0xB1>(x)); x = fmaxf(x, dpp_mov<0x4E>(x)); x = fmaxf(x, dpp_mov<0x141>(x)); x = fmaxf(x, dpp_mov<0x140>(x));
    return fmaxf(fmaxf(rl(x, 0), rl(x, 16)), fmaxf(rl(x, 32), rl(x, 48)));
}
__device__ __forceinline__ int wave_min_dpp(int x) {
    x = min(x, dpp_movi<0xB1>(x)); x = min(x, dpp_movi<0x4E>(x)); x = min(x, dpp_movi<0x141>(x)); x = min(x, dpp_movi<0x140>(x));
    return min(min(__builtin_amdgcn_readlane(x, 0), __builtin_amdgcn_readlane(x, 16)),
               min(__builtin_amdgcn_readlane(x, 32), __builtin_amdgcn_readlane(x, 48)));
}

template <typename TI, int NWV, int PPL>
__global__ __launch_bounds__(NWV * 64) void sinkhorn_regress_kernel(const TI* __restrict__ cv, float* __restrict__ disp,
                                                                    float* __restrict__ conf, float* __restrict__ occ,
                                                                    int32_t* __restrict__ amax, int w, int ot_iter, int use_pos) {
    constexpr int VEC = 16 / sizeof(TI);
    constexpr int EPL = PPL * VEC;                         // columns per lane: piece p covers j = (lane + 64*p)*VEC .. +VEC-1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = w + 1;                                   // padded size
    float* u = reinterpret_cast<float*>(smem);             // [n]
    float* v = u + n;                                      // [n]
    float* pm = v + n;                                     // [NWV][n]  per-wave column partial max
    float* pz = pm + NWV * n;                              // [NWV][n]  per-wave column partial sum

    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TI* S = cv + (size_t)blockIdx.x * w * w;
    const float log_row = -__logf(2.0f * w);               // log(1/(2w))   marginal of a regular row/column
    const float log_bin = __logf(0.5f);                    // log(w/(2w))   marginal of the dustbin
    const float log2w = __logf(2.0f * w);
    float* od = disp + (size_t)blockIdx.x * w;
    float* oc = conf + (size_t)blockIdx.x * w;
    float* oo = occ + (size_t)blockIdx.x * w;

    // a row is requested one iteration ahead as raw 16-byte pieces (no use of the data -> no wait) and decoded when it is consumed
    auto fetch_row = [&](int i, raw16_t (&raw)[PPL]) __attribute__((always_inline)) {
        const TI* Si = S + (size_t)(i < w ? i : 0) * w;
        const int jend = i < w ? (use_pos ? i + 1 : w) : 0;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            const int j0 = (lane + 64 * p) * VEC;
            if (j0 < jend) raw[p] = global_load16(Si + j0);      // w % VEC == 0: a piece starting inside [0, w) is whole
        }
    };
    auto decode_row = [&](int i, const raw16_t (&raw)[PPL], float (&x)[EPL]) __attribute__((always_inline)) {
        const int jend = use_pos ? i + 1 : w;
#pragma unroll
        for (int p = 0; p < PPL; ++p) {
            const int j0 = (lane + 64 * p) * VEC;
            const Vec16<TI> r = __builtin_bit_cast(Vec16<TI>, raw[p]);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                if (i < w) x[p * VEC + e] = (j0 + e < jend) ? to_f32(r.v[e]) : -INFINITY;     // masked triangle / past the row
                else x[p * VEC + e] = (j0 + e < w) ? 0.f : -INFINITY;                          // dustbin row: S = 0, never masked
            }
        }
    };

    LSE col[EPL];                                          // column accumulators of this lane (for the next v)
    LSE bin;                                               // dustbin column (lane-uniform)
    float vreg[EPL];                                       // v_j of this lane's columns for the current row sweep

    for (int pass = 0; pass <= ot_iter; ++pass) {
        const bool last = pass == ot_iter;
#pragma unroll
        for (int c = 0; c < EPL; ++c) col[c].init();
        bin.init();
        if (pass > 0) {
#pragma unroll
            for (int p = 0; p < PPL; ++p)
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const int j = (lane + 64 * p) * VEC + e;
                    vreg[p * VEC + e] = j < w ? v[j] : 0.f;
                }
        }
        const float vbin = pass > 0 ? v[w] : 0.f;
        // rows i = wv, wv + NWV, ... ; the dustbin row i = w (S = 0, never masked) is row number w of the same sequence
        raw16_t rcur[PPL], rnext[PPL];
#pragma unroll
        for (int p = 0; p < PPL; ++p) { rcur[p] = (raw16_t){0.f, 0.f, 0.f, 0.f}; rnext[p] = rcur[p]; }
        fetch_row(wv, rcur);
        for (int i = wv; i <= w; i += NWV) {
            if (i + NWV <= w) fetch_row(i + NWV, rnext);          // in flight under this row's arithmetic
            float x[EPL];
            decode_row(i, rcur, x);
#pragma unroll
            for (int p = 0; p < PPL; ++p) rcur[p] = rnext[p];
            float ui = 0.f;
            if (pass > 0) {
                // ---- row sweep: u_i = log mu_i - LSE_j(S_ij + v_j), dustbin column included (S = 0)
                float mloc = -INFINITY;
#pragma unroll
                for (int c = 0; c < EPL; ++c) mloc = fmaxf(mloc, x[c] + vreg[c]);
                const float m = fmaxf(wave_max_dpp(mloc), vbin);
                float sloc = 0.f;
#pragma unroll
                for (int c = 0; c < EPL; ++c) sloc += __expf(x[c] + vreg[c] - m);
                const float ssum = wave_sum_dpp(sloc) + __expf(vbin - m);
                ui = (i == w ? log_bin : log_row) - (m + __logf(fmaxf(ssum, 1e-30f)));
                if (lane == 0) u[i] = ui;
            }
            if (!last) {
                // ---- column sweep contribution of this row: S_ij + u_i
#pragma unroll
                for (int c = 0; c < EPL; ++c) col[c].add(x[c] + ui);
                bin.add(ui);
            } else if (i < w) {
                // ---- probabilities, argmax (first max wins), window regression, row mass
                const float ci = ui + log2w;
                float best = -1.f, mass = 0.f;
                int bj = 0x7fffffff;
#pragma unroll
                for (int c = 0; c < EPL; ++c) {
                    const float pr = __expf(x[c] + ci + vreg[c]);                  // exp(-inf) = 0 for masked / out-of-range columns
                    mass += pr;
                    const int j = (lane + 64 * (c / VEC)) * VEC + (c % VEC);
                    if (x[c] != -INFINITY && pr > best) { best = pr; bj = j; }    // strict: keeps the first maximum of this lane
                }
                const float bmax = wave_max_dpp(best);
                bj = wave_min_dpp(best == bmax ? bj : 0x7fffffff);
                mass = wave_sum_dpp(mass);
                // 5 taps around the argmax, evaluated by lanes 0..4 (zero outside [0,w) and in the masked triangle)
                const TI* Si = S + (size_t)i * w;
                const int jend = use_pos ? i + 1 : w;
                const int jj = bj + lane - 2;
                float pk = 0.f;
                if (lane < 5 && jj >= 0 && jj < jend) pk = __expf(to_f32(Si[jj]) + ci + v[jj]);
                float cf = 0.f, num = 0.f;
#pragma unroll
                for (int k = 0; k < 5; ++k) {                                     // same summation order as the reference loop
                    const float pr = rl(pk, k);
                    cf += pr;
                    num += pr * (float)(bj + k - 2);
                }
                if (lane == 0) {
                    const float corr = (num + 1e-4f) / (cf + 1e-4f);
                    od[i] = (float)i - corr;
                    oc[i] = cf;
                    oo[i] = mass;
                    if (amax) amax[(size_t)blockIdx.x * w + i] = bj;
                }
            }
        }
        if (last) break;
        // ---- combine the per-wave column partials into v
#pragma unroll
        for (int p = 0; p < PPL; ++p)
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int j = (lane + 64 * p) * VEC + e;
                if (j < w) { pm[wv * n + j] = col[p * VEC + e].m; pz[wv * n + j] = col[p * VEC + e].z; }
            }
        if (lane == 0) { pm[wv * n + w] = bin.m; pz[wv * n + w] = bin.z; }
        __syncthreads();
        for (int j = tid; j < n; j += NWV * 64) {
            LSE t; t.init();
#pragma unroll
            for (int k = 0; k < NWV; ++k) t.merge(pm[k * n + j], pz[k * n + j]);
            v[j] = (j == w ? log_bin : log_row) - t.value();
        }
        __syncthreads();
    }
}

template <typename TI, int PPL>
static int launch_sinkhorn(const void* cv, float* disp, float* conf, float* occ, int32_t* amax, int rows, int w, int ot_iter,
                           int use_pos, hipStream_t st) {
    constexpr int NWV = 16;
    auto kern = sinkhorn_regress_kernel<TI, NWV, PPL>;
    const size_t lds = (size_t)(2 + 2 * NWV) * (w + 1) * sizeof(float);
    static size_t attr_bytes_dev[kMaxDevices] = {};
    size_t& attr_bytes = attr_bytes_dev[current_device()];
    if (lds > attr_bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return set_error("sinkhorn: cannot reserve %zu bytes of LDS", lds);
        attr_bytes = lds;
    }
    hipLaunchKernelGGL(kern, dim3(rows), dim3(NWV * 64), lds, st, static_cast<const TI*>(cv), disp, conf, occ, amax, w, ot_iter, use_pos);
    return check_launch("sinkhorn_regress");
}

template <typename TI>
static int dispatch_ppl(const void* cv, float* disp, float* conf, float* occ, int32_t* amax, int rows, int w, int ot_iter,
                        int use_pos, hipStream_t st) {
    constexpr int VEC = 16 / (int)sizeof(TI);
    const int need = (w + 64 * VEC - 1) / (64 * VEC);          // 16-byte pieces per lane
    if (need <= 1) return launch_sinkhorn<TI, 1>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, st);
    if (need <= 2) return launch_sinkhorn<TI, 2>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, st);
    if (need <= 3) return launch_sinkhorn<TI, 3>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, st);
    if (need <= 4) return launch_sinkhorn<TI, 4>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, st);
    return set_error("sinkhorn: w=%d too large (max %d)", w, 4 * 64 * VEC);
}

}  // namespace s2m2

extern "C" size_t s2m2_sinkhorn_workspace_bytes(int B, int h, int w, int cv_dtype) {
    (void)B; (void)h; (void)w; (void)cv_dtype;
    return 0;                                    // u, v and the partials live in LDS
}

extern "C" int s2m2_sinkhorn_regress(const void* cv, float* disp, float* conf, float* occ, int32_t* argmax, int B, int h, int w,
                                     int ot_iter, int use_positivity, int cv_dtype, void* workspace, void* stream) {
    using namespace s2m2;
    (void)workspace;
    S2M2_REQUIRE(cv && disp && conf && occ, "sinkhorn: null pointer");
    S2M2_REQUIRE(B > 0 && h > 0 && w > 0 && ot_iter >= 1, "sinkhorn: bad arguments B=%d h=%d w=%d ot_iter=%d", B, h, w, ot_iter);
    S2M2_REQUIRE(w % 8 == 0, "sinkhorn: w=%d must be a multiple of 8 (image width multiple of 32)", w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cv_dtype == S2M2_F16) return dispatch_ppl<half_t>(cv, disp, conf, occ, argmax, B * h, w, ot_iter, use_positivity, st);
    if (cv_dtype == S2M2_F32) return dispatch_ppl<float>(cv, disp, conf, occ, argmax, B * h, w, ot_iter, use_positivity, st);
    return set_error("sinkhorn: unsupported cv dtype %d", cv_dtype);
}
