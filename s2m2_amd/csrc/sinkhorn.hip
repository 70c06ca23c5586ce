// K2 -- Sinkhorn optimal transport with dustbins + argmax + 5-tap window regression, one workgroup per image row.
//
// Replaces DispInit._optimal_transport/_sinkhorn (log-domain, ot_iter sweeps of logsumexp_stable), the
// probability recovery, argmax and the gather-based window expectation
// (/root/reference/src/s2m2/core/model/submodules.py:147-152,169-201,211-241; SURVEY.md A5+A6, Appendix A 3-10).
//
// For one (b,y) the score matrix S (w x w, + one dustbin row and column of zeros) is swept 2*ot_iter times:
//     v_j = log nu_j - LSE_i(S_ij + u_i)      (column sweep; first one with u = 0)
//     u_i = log mu_i - LSE_j(S_ij + v_j)      (row sweep)
// then P_ij = exp(S_ij + u_i + v_j + log 2w) for i,j < w.  With use_positivity the reference fills j > i with
// -1e4, which underflows to exactly 0 after exp in fp32, so those entries are skipped (bit-identical).
//
// The reference materialises ~35 full-volume temporaries; here S is only ever READ (from L2 / Infinity Cache
// after the first touch), u and v live in LDS, and each sweep keeps an online (max, sum) pair per thread with a
// lazy rescale, i.e. ~1 exp per element instead of max-pass + exp-pass.  All waves of the block own rows
// i = wave, wave+NWV, ... so that both sweeps read S row-contiguously (coalesced along j).
#include "common.h"

namespace s2m2 {

struct LSE {                         // running log-sum-exp state: sum of exp(x - m)
    float m, z;
    __device__ __forceinline__ void init() { m = -INFINITY; z = 0.f; }
    __device__ __forceinline__ void add(float x) {
        if (x > m) { z = z * __expf(m - x) + 1.0f; m = x; }     // exp(-inf) = 0 on the first element
        else       { z += __expf(x - m); }
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

template <typename TI, int NWV, int CPL>
__global__ __launch_bounds__(NWV * 64) void sinkhorn_regress_kernel(const TI* __restrict__ cv, float* __restrict__ disp,
                                                                    float* __restrict__ conf, float* __restrict__ occ,
                                                                    int32_t* __restrict__ amax, int w, int ot_iter, int use_pos) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = w + 1;                                   // padded size
    float* u = reinterpret_cast<float*>(smem);             // [n]
    float* v = u + n;                                      // [n]
    float* pm = v + n;                                     // [NWV][n]  per-wave column partial max
    float* pz = pm + NWV * n;                              // [NWV][n]  per-wave column partial sum

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const TI* S = cv + (size_t)blockIdx.x * w * w;
    const float log_row = -__logf(2.0f * w);               // log(1/(2w))   marginal of a regular row/column
    const float log_bin = __logf(0.5f);                    // log(w/(2w))   marginal of the dustbin

    for (int it = 0; it < ot_iter; ++it) {
        // ---------------- column sweep: v_j ----------------
        LSE col[CPL];
#pragma unroll
        for (int m = 0; m < CPL; ++m) col[m].init();
        for (int i = wv; i < n; i += NWV) {
            const float ui = (it == 0) ? 0.f : u[i];
            const TI* Si = S + (size_t)i * w;
#pragma unroll
            for (int m = 0; m < CPL; ++m) {
                const int j = lane + 64 * m;
                if (j < n) {
                    if (i == w || j == w) col[m].add(ui);                               // dustbin row / column: S = 0
                    else if (!use_pos || j <= i) col[m].add(to_f32(Si[j]) + ui);
                }
            }
        }
#pragma unroll
        for (int m = 0; m < CPL; ++m) {
            const int j = lane + 64 * m;
            if (j < n) { pm[wv * n + j] = col[m].m; pz[wv * n + j] = col[m].z; }
        }
        __syncthreads();
        for (int j = tid; j < n; j += NWV * 64) {
            LSE t; t.init();
#pragma unroll
            for (int k = 0; k < NWV; ++k) t.merge(pm[k * n + j], pz[k * n + j]);
            v[j] = (j == w ? log_bin : log_row) - t.value();
        }
        __syncthreads();
        // ---------------- row sweep: u_i ----------------
        for (int i = wv; i < n; i += NWV) {
            LSE r; r.init();
            const TI* Si = S + (size_t)i * w;
            const int jend = (i == w) ? w : (use_pos ? i + 1 : w);                      // valid j in [0, jend)
            for (int j = lane; j < jend; j += 64) r.add((i == w ? 0.f : to_f32(Si[j])) + v[j]);
            if (lane == 0) r.add(v[w]);                                                 // dustbin column
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r.merge(__shfl_xor(r.m, o, 64), __shfl_xor(r.z, o, 64));
            if (lane == 0) u[i] = (i == w ? log_bin : log_row) - r.value();
        }
        __syncthreads();
    }

    // ---------------- probabilities, argmax (first max wins), window regression, row mass ----------------
    const float log2w = __logf(2.0f * w);
    float* od = disp + (size_t)blockIdx.x * w;
    float* oc = conf + (size_t)blockIdx.x * w;
    float* oo = occ + (size_t)blockIdx.x * w;
    for (int i = wv; i < w; i += NWV) {
        const TI* Si = S + (size_t)i * w;
        const float ci = u[i] + log2w;
        const int jend = use_pos ? i + 1 : w;
        float best = -1.f, mass = 0.f;
        int bj = 0x7fffffff;
        for (int j = lane; j < jend; j += 64) {
            const float p = __expf(to_f32(Si[j]) + ci + v[j]);
            mass += p;
            if (p > best) { best = p; bj = j; }                   // strict: keeps the first maximum of this lane
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oj = __shfl_xor(bj, o, 64);
            if (ob > best || (ob == best && oj < bj)) { best = ob; bj = oj; }
            mass += __shfl_xor(mass, o, 64);
        }
        // 5 taps around the argmax, evaluated by lanes 0..4 (zero outside [0,w) and in the masked triangle)
        const int jj = bj + lane - 2;
        float pk = 0.f;
        if (lane < 5 && jj >= 0 && jj < jend) pk = __expf(to_f32(Si[jj]) + ci + v[jj]);
        float cf = 0.f, num = 0.f;
#pragma unroll
        for (int k = 0; k < 5; ++k) {                             // same summation order as the reference loop
            const float p = __shfl(pk, k, 64);
            cf += p;
            num += p * (float)(bj + k - 2);
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

template <typename TI, int CPL>
static int launch_sinkhorn(const void* cv, float* disp, float* conf, float* occ, int32_t* amax, int rows, int w, int ot_iter,
                           int use_pos, hipStream_t st) {
    constexpr int NWV = 16;
    auto kern = sinkhorn_regress_kernel<TI, NWV, CPL>;
    const size_t lds = (size_t)(2 + 2 * NWV) * (w + 1) * sizeof(float);
    static size_t attr_bytes = 0;
    if (lds > attr_bytes) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return set_error("sinkhorn: cannot reserve %zu bytes of LDS", lds);
        attr_bytes = lds;
    }
    hipLaunchKernelGGL(kern, dim3(rows), dim3(NWV * 64), lds, st, static_cast<const TI*>(cv), disp, conf, occ, amax, w, ot_iter, use_pos);
    return check_launch("sinkhorn_regress");
}

template <typename TI>
static int dispatch_cpl(const void* cv, float* disp, float* conf, float* occ, int32_t* amax, int rows, int w, int ot_iter,
                        int use_pos, hipStream_t st) {
    const int need = (w + 1 + 63) / 64;
    if (need <= 3) return launch_sinkhorn<TI, 3>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, st);
    if (need <= 5) return launch_sinkhorn<TI, 5>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, st);
    if (need <= 10) return launch_sinkhorn<TI, 10>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, st);
    if (need <= 17) return launch_sinkhorn<TI, 17>(cv, disp, conf, occ, amax, rows, w, ot_iter, use_pos, st);
    return set_error("sinkhorn: w=%d too large (max 1087)", w);
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
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cv_dtype == S2M2_F16) return dispatch_cpl<half_t>(cv, disp, conf, occ, argmax, B * h, w, ot_iter, use_positivity, st);
    if (cv_dtype == S2M2_F32) return dispatch_cpl<float>(cv, disp, conf, occ, argmax, B * h, w, ot_iter, use_positivity, st);
    return set_error("sinkhorn: unsupported cv dtype %d", cv_dtype);
}
