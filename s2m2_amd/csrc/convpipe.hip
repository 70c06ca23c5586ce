// EXPERIMENT (round 6): the fragment-stream convolution (K5 v5, conv.hip) with its tile loads taken off the compute waves.
//
// v5 loads the patch + halo of a 128-channel chunk through registers into LDS, meets at a barrier, runs the K loop, and meets again for the next
// chunk: in a single-round launch every block of the chip loads at the same time and multiplies at the same time (profiles/r05/frag_span.txt: 5.8 us
// of a 29 us block life are the tile load; profiles/r06/pmc_forward.txt: the matrix pipe is busy 0.33 - 0.44 of the cycles).  Here
//   * a block is 4 compute waves (one per 32 couts, all pixels of the patch: the v5 decomposition) + ONE LOAD WAVE that does nothing but LDS-DMA
//     (global_load_lds, 16 bytes per lane, no registers): the compute waves' vmcnt stream holds their weight ring and nothing else, so the counted
//     waits of the ring never wait for a tile load;
//   * the channels come in chunks of 64 through two LDS buffers: the DMA of chunk 1 runs under the K loop of chunk 0, the DMA of the NEXT patch's
//     chunk 0 under the K loop of chunk 1 and the epilogue (blocks are persistent: patch b, b + gridDim.x, ...);
//   * two barriers per patch, joined by all five waves (A: chunk 0 has landed / the buffers of the previous patch are free; B: chunk 1 has landed /
//     chunk 0 is consumed); the epilogue stores straight from the accumulators (8 bytes per lane and 4 couts: no staging tile, no barrier).
// K order of the weight stream: (chunk of 64, tap, k16 step) -- packed by the caller (tools/convpipe_bench.py: the torch formula of pack.pack_conv_frag
// with CK = 64).  fp16, ONE source tensor, Cin = 128, Cout a multiple of 128, 3x3, stride 1, epilogue: bias + NONE / GELU / RELU, optional residual add.
// Entry point s2m2_debug_conv_pipe: not part of the C ABI of include/s2m2_hip.h (an experiment's bench hook, like s2m2_debug_frag_trace).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "common.h"
#include "epilogue.h"

namespace s2m2 {

struct PipeArgs {
    const half_t* x; long long xs;
    half_t* out; long long os;
    const half_t* aux; long long as;
    const raw16_t* w;
    const float* bias;
    const void* zero;
    int N, H, W, Cout, act;
    int tiles_x, tiles_y, npatch;
};

template <int PW_>
struct PipeCfg {
    static constexpr int PW = PW_, PH = 4, BM = PW * PH, MT = BM / 32;
    static constexpr int HW = PW + 2, HH = PH + 2, NPX = HW * HH;
    static constexpr int CH = 64, KS = CH / 16, RS = CH + 8;          // halfs per halo pixel in LDS: 8 pieces + one of padding
    static constexpr int UPP = RS / 8;                                 // 16-byte units per pixel (9)
    static constexpr int UNITS = NPX * UPP, NDMA = (UNITS + 63) / 64;
    static constexpr size_t BUF_BYTES = (size_t)NDMA * 1024;
    static constexpr int NTAP = 9, NCHUNK = 2, NFRAG = NCHUNK * NTAP * KS;
    static_assert(BM % 32 == 0, "whole MFMA pixel tiles");
};

__device__ __forceinline__ void pipe_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename CFG, int ACT, bool ADD>
__global__ __launch_bounds__(320, 3) void conv_pipe_kernel(PipeArgs p) {
    constexpr int PW = CFG::PW, PH = CFG::PH, MT = CFG::MT, HW = CFG::HW, RS = CFG::RS, KS = CFG::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* B0 = reinterpret_cast<half_t*>(smem);
    half_t* B1 = reinterpret_cast<half_t*>(smem + CFG::BUF_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int n0 = blockIdx.y * 128;

    if (wv == 4) {
        // ------------------------------------------------------------------ the load wave
        // unit U = 64 k + lane of a chunk tile: halo pixel U / 9 = (hy, hx), piece U % 9 (piece 8: the padding of the pixel's LDS row)
        auto issue = [&](int pid, int chunk, half_t* buf) __attribute__((always_inline)) {
            int b = pid;
            const int tx = b % p.tiles_x; b /= p.tiles_x;
            const int ty = b % p.tiles_y;
            const int n = b / p.tiles_y;
            const int y0 = ty * PH - 1, x0 = tx * PW - 1;
            const half_t* base = p.x + (long long)n * p.H * p.W * p.xs + chunk * CFG::CH;
#pragma unroll 2
            for (int k = 0; k < CFG::NDMA; ++k) {
                const int U = k * 64 + lane, px = U / CFG::UPP, q = U - px * CFG::UPP;      // (constant divisors: multiply-shift)
                const int hy = px / HW, hx = px - hy * HW;
                const int yy = y0 + hy, xx = x0 + hx;
                const bool ok = px < CFG::NPX && q < 8 && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                const half_t* src = ok ? base + ((long long)yy * p.W + xx) * p.xs + q * 8 : static_cast<const half_t*>(p.zero);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(buf + k * 512), 16, 0, 0);
            }
        };
        int pid = blockIdx.x;
        if (pid < p.npatch) issue(pid, 0, B0);
        for (; pid < p.npatch; pid += gridDim.x) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pipe_barrier();                                       // A: chunk 0 of this patch is in B0; B1 is free
            issue(pid, 1, B1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pipe_barrier();                                       // B: chunk 1 is in B1; B0 is consumed
            if (pid + (int)gridDim.x < p.npatch) issue(pid + gridDim.x, 0, B0);
        }
        return;
    }

    // ---------------------------------------------------------------------- the compute waves: wave wv owns couts n0 + 32 wv .. + 31 of every pixel
    const raw16_t* wf = p.w + ((size_t)(blockIdx.y * 4 + wv) * CFG::NFRAG) * 64 + lane;
    int poff[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int q = 32 * i + l31, qy = q / PW, qx = q - qy * PW;
        poff[i] = (qy * HW + qx) * RS + hi * 8;
    }
    raw16_t ring[8];
    for (int pid = blockIdx.x; pid < p.npatch; pid += gridDim.x) {
        float16_t acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        // fragments 0 .. 6 of the stream (slot 7 is requested by step 0)
#pragma unroll
        for (int s = 0; s < 7; ++s) global_load16_async(ring[s], wf + (size_t)s * 64);
        int g = 0;                                                // k16 step of the patch = fragment consumed next
        // the KS k16 steps of one tap; ring slots PAR * 4 .. + 3.  ONE set of pixel fragments (MT x 4 registers): the fragments of the next step --
        // at the end of a tap: of the next tap's first step, `anext` -- are requested right behind this step's MFMAs and arrive under them
        // (two sets, as in conv.hip, do not fit 168 registers next to MT = 5 accumulator tiles and the ring)
        Frag<half_t> xf[MT];
        auto tap = [&](auto par_tag, const half_t* a, const half_t* anext) __attribute__((always_inline)) {
            constexpr int PAR = decltype(par_tag)::value;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                wait_vmcnt<6>();                                  // the fragment of this step was requested 7 requests ago
                settle(ring[PAR * 4 + kk]);
                Frag<half_t> wfr;
                wfr.v = __builtin_bit_cast(half8_t, ring[PAR * 4 + kk]);
#pragma unroll
                for (int i = 0; i < MT; ++i) mma32(acc[i], wfr, xf[i]);
                const half_t* nx = kk + 1 < KS ? a + (kk + 1) * 16 : anext;
#pragma unroll
                for (int i = 0; i < MT; ++i) load_frag(xf[i], nx + poff[i]);
                {
                    const int f = g + 7;                          // UNCONDITIONAL refill of the slot consumed one step ago (see conv.hip)
                    global_load16_async(ring[(PAR * 4 + kk + 7) % 8], wf + (size_t)(f < CFG::NFRAG ? f : CFG::NFRAG - 1) * 64);
                }
                ++g;
            }
        };
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        auto tap_at = [&](const half_t* buf, int t) { return buf + ((t / 3) * HW + (t % 3)) * RS; };      // tap t = (ky, kx) = (t / 3, t % 3)
        pipe_barrier();                                           // A
#pragma unroll
        for (int i = 0; i < MT; ++i) load_frag(xf[i], B0 + poff[i]);
        {   // chunk 0: taps 0 .. 8 on ring halves 0 1 0 1 .. 0
#pragma unroll 1
            for (int t = 0; t < 8; t += 2) {
                tap(P0{}, tap_at(B0, t), tap_at(B0, t + 1));
                tap(P1{}, tap_at(B0, t + 1), tap_at(B0, t + 2));
            }
            tap(P0{}, tap_at(B0, 8), tap_at(B0, 8));              // (the fragments requested at its end are never used: chunk 1 is not there yet)
        }
        pipe_barrier();                                           // B
#pragma unroll
        for (int i = 0; i < MT; ++i) load_frag(xf[i], B1 + poff[i]);
        {   // chunk 1: taps 0 .. 8 on ring halves 1 0 1 0 .. 1
            tap(P1{}, tap_at(B1, 0), tap_at(B1, 1));
#pragma unroll 1
            for (int t = 1; t < 9; t += 2) {
                tap(P0{}, tap_at(B1, t), tap_at(B1, t + 1));
                tap(P1{}, tap_at(B1, t + 1), tap_at(B1, t + 2 < 9 ? t + 2 : 8));
            }
        }
        wait_vmcnt<0>();                                          // the tail requests land before their registers are reused
#pragma unroll
        for (int s = 0; s < 8; ++s) settle(ring[s]);

        float4_t bias4[4];                                        // (read here, not held through the K loop: 16 registers)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) bias4[g4] = *reinterpret_cast<const float4_t*>(p.bias + n0 + wv * 32 + 8 * g4 + 4 * hi);
        // ---- epilogue, straight from the accumulators: lane (pixel l31 of tile i, half hi) holds couts 8 g + 4 hi + e, g < 4, e < 4
        int b = pid;
        const int tx = b % p.tiles_x; b /= p.tiles_x;
        const int ty = b % p.tiles_y;
        const int n = b / p.tiles_y;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int q = 32 * i + l31, qy = q / PW, qx = q - qy * PW;
            const int yy = ty * PH + qy, xx = tx * PW + qx;
            const bool ok = yy < p.H && xx < p.W;
            const long long pix = ((long long)n * p.H + yy) * p.W + xx;
            half_t* orow = p.out + pix * p.os + n0 + wv * 32 + 4 * hi;
            const half_t* arow = ADD ? p.aux + pix * p.as + n0 + wv * 32 + 4 * hi : nullptr;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = acc[i][4 * g4 + e] + bias4[g4][e];
                half4_t o;
                if constexpr (ACT == S2M2_ACT_GELU) {
                    const float2_t r0 = fast_gelu16x2((float2_t){v[0], v[1]}), r1 = fast_gelu16x2((float2_t){v[2], v[3]});
                    o = half4_t{(half_t)r0.x, (half_t)r0.y, (half_t)r1.x, (half_t)r1.y};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = from_f32<half_t>(activate_to<ACT, half_t>(v[e]));
                }
                if constexpr (ADD) {
                    if (ok) {
                        const half4_t u = *reinterpret_cast<const half4_t*>(arow + 8 * g4);
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = from_f32<half_t>((float)o[e] + (float)u[e]);
                    }
                }
                if (ok) *reinterpret_cast<half4_t*>(orow + 8 * g4) = o;
            }
        }
    }
}

template <int PW, int ACT, bool ADD>
static int launch_pipe(PipeArgs a, int Cout, hipStream_t st) {
    using CFG = PipeCfg<PW>;
    a.tiles_x = (a.W + PW - 1) / PW;
    a.tiles_y = (a.H + 3) / 4;
    a.npatch = a.N * a.tiles_x * a.tiles_y;
    auto kern = conv_pipe_kernel<CFG, ACT, ADD>;
    static size_t granted[kMaxDevices] = {};
    if (reserve_lds(reinterpret_cast<const void*>(kern), 2 * CFG::BUF_BYTES, granted, "conv_pipe")) return 1;
    const int grid = a.npatch < 512 ? a.npatch : 512;
    hipLaunchKernelGGL(kern, dim3(grid, Cout / 128), dim3(320), 2 * CFG::BUF_BYTES, st, a);
    return check_launch("conv_pipe");
}

}  // namespace s2m2

extern "C" int s2m2_debug_conv_pipe(const void* x, long long x_stride, void* out, long long out_stride, const void* aux, long long aux_stride,
                                    int N, int H, int W, int Cin, int Cout, const void* w_frag64, const float* bias, int act, int pw, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(x && out && w_frag64 && bias, "conv_pipe: null pointer");
    S2M2_REQUIRE(Cin == 128 && Cout > 0 && Cout % 128 == 0 && N > 0 && H > 0 && W > 0, "conv_pipe: Cin = 128, Cout a multiple of 128");
    S2M2_REQUIRE(x_stride >= Cin && out_stride >= Cout && x_stride % 8 == 0 && out_stride % 4 == 0 && (!aux || (aux_stride >= Cout && aux_stride % 4 == 0)),
                 "conv_pipe: strides");
    S2M2_REQUIRE(act == S2M2_ACT_NONE || act == S2M2_ACT_GELU || act == S2M2_ACT_RELU, "conv_pipe: act");
    S2M2_REQUIRE(pw == 32 || pw == 40, "conv_pipe: pw = 32 or 40");
    PipeArgs a;
    a.x = static_cast<const half_t*>(x); a.xs = x_stride; a.out = static_cast<half_t*>(out); a.os = out_stride;
    a.aux = static_cast<const half_t*>(aux); a.as = aux_stride;
    a.w = static_cast<const raw16_t*>(w_frag64); a.bias = bias; a.N = N; a.H = H; a.W = W; a.Cout = Cout; a.act = act;
    a.zero = zero_page();
    S2M2_REQUIRE(a.zero, "conv_pipe: cannot allocate the zero page");
    hipStream_t st = static_cast<hipStream_t>(stream);
#define S2M2_PIPE(PWV)                                                                                                   \
    if (aux) {                                                                                                           \
        if (act == S2M2_ACT_GELU) return launch_pipe<PWV, S2M2_ACT_GELU, true>(a, Cout, st);                             \
        if (act == S2M2_ACT_RELU) return launch_pipe<PWV, S2M2_ACT_RELU, true>(a, Cout, st);                             \
        return launch_pipe<PWV, S2M2_ACT_NONE, true>(a, Cout, st);                                                       \
    } else {                                                                                                             \
        if (act == S2M2_ACT_GELU) return launch_pipe<PWV, S2M2_ACT_GELU, false>(a, Cout, st);                            \
        if (act == S2M2_ACT_RELU) return launch_pipe<PWV, S2M2_ACT_RELU, false>(a, Cout, st);                            \
        return launch_pipe<PWV, S2M2_ACT_NONE, false>(a, Cout, st);                                                      \
    }
    if (pw == 40) { S2M2_PIPE(40) }
    S2M2_PIPE(32)
#undef S2M2_PIPE
}
