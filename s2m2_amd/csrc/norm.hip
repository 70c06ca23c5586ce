// K6 -- normalisation kernels (HBM-bound, one read + one write of the activation).
//
//  * layernorm_rows: pre-norm LayerNorm WITHOUT affine over the channel axis of token rows
//    (reference attentions.py:117,148,182,213,243: nn.LayerNorm(dim, elementwise_affine=False), eps 1e-5, biased variance).
//  * groupnorm_nhwc: nn.GroupNorm(G, C) with affine on an NHWC activation (reference submodules.py:80,90 -- the one
//    normalisation of the CNN backbone; PyTorch's channels-last GroupNorm takes 4.6 ms here, see profiles/r01).
//    Two launches: per-(sample, group) sum / sum of squares accumulated in fp64 (block partials in fp32, one fp64
//    atomicAdd pair per block and group), then the normalise+affine sweep.
// Both compute in fp32 regardless of the I/O dtype.
#include "common.h"
#include <stdlib.h>
#include "plan.h"

namespace s2m2 {

// 16 lanes per token row, 4 rows per wave; each lane owns 16-byte pieces lane16, lane16+16, ...
template <typename T, int MAXP>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const T* __restrict__ x, T* __restrict__ y, long long rows, int C,
                                                             long long xs, long long ys) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane16 = threadIdx.x & 15;
    const long long row = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    if (row >= rows) return;                                      // whole 16-lane groups leave together
    const int P = C / VEC;
    const T* xr = x + row * xs;
    Vec16<T> v[MAXP];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
        const int pi = lane16 + 16 * q;
        if (pi < P) {
            v[q] = *reinterpret_cast<const Vec16<T>*>(xr + pi * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s += to_f32(v[q].v[e]);
        }
    }
    const float mean = group_sum<16>(s) / (float)C;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
        const int pi = lane16 + 16 * q;
        if (pi < P) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = to_f32(v[q].v[e]) - mean; ss += d * d; }
        }
    }
    const float rstd = rsqrtf(group_sum<16>(ss) / (float)C + 1e-5f);
    T* yr = y + row * ys;
#pragma unroll
    for (int q = 0; q < MAXP; ++q) {
        const int pi = lane16 + 16 * q;
        if (pi < P) {
            Vec16<T> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>((to_f32(v[q].v[e]) - mean) * rstd);
            *reinterpret_cast<Vec16<T>*>(yr + pi * VEC) = o;
        }
    }
}

template <typename T, int MAXP>
static int launch_ln(const void* x, void* y, long long rows, int C, long long xs, long long ys, hipStream_t st) {
    const long long threads = rows * 16;
    hipLaunchKernelGGL((layernorm_rows_kernel<T, MAXP>), dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st,
                       static_cast<const T*>(x), static_cast<T*>(y), rows, C, xs, ys);
    return check_launch("layernorm");
}

template <typename T>
static int dispatch_ln(const void* x, void* y, long long rows, int C, long long xs, long long ys, hipStream_t st) {
    const int P = C / (16 / (int)sizeof(T));
    const int need = (P + 15) / 16;
    if (need <= 1) return launch_ln<T, 1>(x, y, rows, C, xs, ys, st);
    if (need <= 2) return launch_ln<T, 2>(x, y, rows, C, xs, ys, st);
    if (need <= 4) return launch_ln<T, 4>(x, y, rows, C, xs, ys, st);
    if (need <= 6) return launch_ln<T, 6>(x, y, rows, C, xs, ys, st);
    if (need <= 12) return launch_ln<T, 12>(x, y, rows, C, xs, ys, st);
    return set_error("layernorm: C=%d too large", C);
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm on NHWC.  stats[(n*G + g)*2 + {0,1}] = sum, sum of squares (fp64).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kGnReplicas = 32;
template <typename T>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const T* __restrict__ x, double* __restrict__ stats, long long HW,
                                                              int C, int G, int pix_per_block) {
    constexpr int VEC = 16 / sizeof(T);
    __shared__ float red[2][32][4];                                // [sum|sq][group][wave]
    const int n = blockIdx.y;
    const int P = C / VEC;                                         // pieces per pixel
    const int cpg = C / G;                                         // channels per group (multiple of VEC)
    const long long p0 = (long long)blockIdx.x * pix_per_block;
    const long long p1 = p0 + pix_per_block < HW ? p0 + pix_per_block : HW;
    const T* xn = x + (long long)n * HW * C;
    // a thread always visits the same piece column -> one group per thread: the (256 / P) * P lowest threads stride by that count
    const int nact = (256 / P) * P;
    const int pi = threadIdx.x % P;
    const int g = pi * VEC / cpg;
    float s = 0.f, ss = 0.f;
    // four pieces in flight per thread (one piece per trip left a single 16-byte request per lane between dependent accumulations: the pass ran at
    // 1.6 TB/s on a tensor that sits in the Infinity Cache)
    const long long qend = p1 * P;
    for (long long q = p0 * P + threadIdx.x; q < qend && (int)threadIdx.x < nact; q += 4LL * nact) {
        Vec16<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long qq = q + (long long)u * nact;
            v[u] = qq < qend ? *reinterpret_cast<const Vec16<T>*>(xn + qq * VEC) : Vec16<T>{};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float f = to_f32(v[u].v[e]); s += f; ss = __builtin_fmaf(f, f, ss); }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ppg = cpg / VEC;                                     // pieces (= consecutive lanes) per group
    if (64 % P == 0 && (ppg & (ppg - 1)) == 0) {
        // lanes of a wave that share a group: same (lane % P) / ppg -- the lanes differing in the bits below ppg and in the bits from P upwards
        for (int o = 1; o < ppg; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
        for (int o = P; o < 64; o <<= 1) { s += __shfl_xor(s, o, 64); ss += __shfl_xor(ss, o, 64); }
        if (lane < P && lane % ppg == 0) { red[0][lane / ppg][wv] = s; red[1][lane / ppg][wv] = ss; }
    } else {
        // general channel counts: one masked wave sum per group
        for (int gg = 0; gg < G; ++gg) {
            const float a = wave_sum(g == gg ? s : 0.f);
            const float b = wave_sum(g == gg ? ss : 0.f);
            if (lane == 0) { red[0][gg][wv] = a; red[1][gg][wv] = b; }
        }
    }
    __syncthreads();
    if (threadIdx.x < G) {
        const float a = red[0][threadIdx.x][0] + red[0][threadIdx.x][1] + red[0][threadIdx.x][2] + red[0][threadIdx.x][3];
        const float b = red[1][threadIdx.x][0] + red[1][threadIdx.x][1] + red[1][threadIdx.x][2] + red[1][threadIdx.x][3];
        // kGnReplicas copies of the 2 N G sums, block b adds to copy b % kGnReplicas: 1216 blocks x 16 fp64 atomics on 32 addresses cost
        // ~17 us of a 46 us pass at (2, 512 x 608, 128) -- the atomics of one address are served one after the other (r06: tools A/B with
        // the pixels per block, 128 / 256 / 512 / 1024 px: 167 / 121 / 96 / 88 us per GroupNorm)
        double* rep = stats + (size_t)(blockIdx.x % kGnReplicas) * 2 * gridDim.y * G;
        atomicAdd(&rep[((long long)n * G + threadIdx.x) * 2 + 0], (double)a);
        atomicAdd(&rep[((long long)n * G + threadIdx.x) * 2 + 1], (double)b);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                              const double* __restrict__ stats, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, long long HW, int C, int G, float eps) {
    constexpr int VEC = 16 / sizeof(T);
    const int n = blockIdx.y;
    const int P = C / VEC, cpg = C / G;
    const long long total = HW * P;
    // a thread always visits the same piece column (the (256 / P) * P lowest threads of a block stride by a multiple of P): its group's
    // mean / rstd (fp64 statistics -> two fp64 divisions) and its 2 * VEC affine coefficients are formed ONCE, not per 16-byte piece --
    // per piece they cost more than the memory traffic of the pass (round 4: 117 -> see profiles/r04/layer_trace_eager.txt)
    const int nact = (256 / P) * P;
    if ((int)threadIdx.x >= nact) return;
    const int pi = threadIdx.x % P;
    const int c0 = pi * VEC;
    const int g = c0 / cpg;
    const double cnt = (double)HW * cpg;
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < kGnReplicas; ++r) {                         // (fixed order: the copies themselves are order-independent up to fp64 rounding)
        const double* rep = stats + (size_t)r * 2 * gridDim.y * G;
        s1 += rep[((long long)n * G + g) * 2];
        s2 += rep[((long long)n * G + g) * 2 + 1];
    }
    const double m = s1 / cnt;
    double var = s2 / cnt - m * m;
    var = var > 0 ? var : 0;
    const float mean = (float)m, rstd = rsqrtf((float)var + eps);
    float ga[VEC], be[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { ga[e] = gamma[c0 + e]; be[e] = beta[c0 + e]; }
    const long long stride = (long long)gridDim.x * nact;
    for (long long q = (long long)blockIdx.x * nact + threadIdx.x; q < total; q += stride) {
        const long long off = ((long long)n * HW * P + q) * VEC;
        const Vec16<T> v = *reinterpret_cast<const Vec16<T>*>(x + off);
        Vec16<T> o;
#pragma unroll
        for (int e = 0; e < VEC; ++e) o.v[e] = from_f32<T>((to_f32(v.v[e]) - mean) * rstd * ga[e] + be[e]);
        *reinterpret_cast<Vec16<T>*>(y + off) = o;
    }
}

template <typename T>
static int run_groupnorm(const void* x, void* y, const float* gamma, const float* beta, double* ws, int N, long long HW, int C, int G,
                         float eps, hipStream_t st) {
    constexpr int VEC = 16 / sizeof(T);
    const int P = C / VEC;
    if (hipMemsetAsync(ws, 0, sizeof(double) * 2 * N * G * kGnReplicas, st) != hipSuccess) return set_error("groupnorm: memset failed");
    static const int ppb_env = getenv("S2M2_GN_PPB") ? atoi(getenv("S2M2_GN_PPB")) : 0;      // tuning only
    const int ppb = ppb_env > 0 ? ppb_env : 512;                   // pixels per block of the statistics pass
    dim3 g1((unsigned)((HW + ppb - 1) / ppb), N);
    hipLaunchKernelGGL((groupnorm_stats_kernel<T>), g1, dim3(256), 0, st, static_cast<const T*>(x), ws, HW, C, G, ppb);
    if (int rc = check_launch("groupnorm_stats")) return rc;
    long long blocks = (HW * P + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL((groupnorm_apply_kernel<T>), dim3((unsigned)blocks, N), dim3(256), 0, st, static_cast<const T*>(x),
                       static_cast<T*>(y), ws, gamma, beta, HW, C, G, eps);
    return check_launch("groupnorm_apply");
}

}  // namespace s2m2

static int layernorm_impl(const void* x, void* y, long long rows, int C, long long x_stride, long long y_stride, int dtype,
                              void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(x && y, "layernorm: null pointer");
    S2M2_REQUIRE(rows > 0 && C > 0 && C % 8 == 0 && x_stride % 8 == 0 && y_stride % 8 == 0,
                 "layernorm: rows=%lld C=%d strides %lld/%lld (C and strides must be multiples of 8)", rows, C, x_stride, y_stride);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == S2M2_F16) return dispatch_ln<half_t>(x, y, rows, C, x_stride, y_stride, st);
    if (dtype == S2M2_F32) return dispatch_ln<float>(x, y, rows, C, x_stride, y_stride, st);
    return set_error("layernorm: unsupported dtype %d", dtype);
}
extern "C" int s2m2_layernorm(const void* x, void* y, long long rows, int C, long long x_stride, long long y_stride, int dtype,
                              void* stream) {
    return s2m2::plan_dispatch("s2m2_layernorm", &layernorm_impl, stream, x, y, rows, C, x_stride, y_stride, dtype);
}


extern "C" size_t s2m2_groupnorm_workspace_bytes(int N, int G) { return sizeof(double) * 2 * (size_t)N * G * s2m2::kGnReplicas; }

static int groupnorm_nhwc_impl(const void* x, void* y, const float* gamma, const float* beta, void* workspace, int N,
                                   long long HW, int C, int G, float eps, int dtype, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(x && y && gamma && beta && workspace, "groupnorm: null pointer");
    const int vec = dtype == S2M2_F16 ? 8 : 4;
    S2M2_REQUIRE(N > 0 && HW > 0 && G > 0 && G <= 32 && C % G == 0 && (C / G) % vec == 0,
                 "groupnorm: N=%d HW=%lld C=%d G=%d (C/G must be a multiple of %d, G <= 32)", N, HW, C, G, vec);
    S2M2_REQUIRE(C / vec <= 256, "groupnorm: C=%d too wide", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == S2M2_F16) return run_groupnorm<half_t>(x, y, gamma, beta, static_cast<double*>(workspace), N, HW, C, G, eps, st);
    if (dtype == S2M2_F32) return run_groupnorm<float>(x, y, gamma, beta, static_cast<double*>(workspace), N, HW, C, G, eps, st);
    return set_error("groupnorm: unsupported dtype %d", dtype);
}
extern "C" int s2m2_groupnorm_nhwc(const void* x, void* y, const float* gamma, const float* beta, void* workspace, int N,
                                   long long HW, int C, int G, float eps, int dtype, void* stream) {
    return s2m2::plan_dispatch("s2m2_groupnorm_nhwc", &groupnorm_nhwc_impl, stream, x, y, gamma, beta, workspace, N, HW, C, G, eps, dtype);
}

