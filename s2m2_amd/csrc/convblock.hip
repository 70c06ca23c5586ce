// K14 -- a whole ConvBlock2D in ONE launch for the coarse pyramid levels (s2m2_conv_block).
//
// Reference: ConvBlock2D.forward (attentions.py:255-281):   out = convs.2(GELU(convs.0(z))) + convs_1x.2(ReLU(convs_1x.0(z)))
// with convs.* 3x3 and convs_1x.* 1x1, all C -> C.  At 1/8 and 1/16 resolution (128 x 152 ... 32 x 38 pixels) the three launches it used to
// be (K9 two-stage chain, K5 v5, K5 v5 with the residual epilogue) are latency chains of 8 - 19 us each whose MFMA phase is 2 - 4 us:
// launch boundary, halo tile, epilogue and store are paid three times and the intermediate tensors make an HBM round trip.  Here
//
//   block = a PH x 32 patch of output pixels x ALL C output channels (one wave per 32 channels);
//   phase 1: convs.0 + bias + GELU on the patch PLUS its one-pixel ring ((PH + 2) x 34 pixels, recomputed by the neighbouring blocks:
//            1.4 - 1.75 x the MFMA work of the unfused layer, which these levels have time for) from a (PH + 4) x 36 input tile in LDS; the
//            result stays in LDS as fp16 (zero outside the image: the second layer's padding).  The same input tile's centre pixels feed
//            convs_1x.0 (accumulated beside phase 1, chunk by chunk);
//   phase 2: convs.2 on the patch from the LDS-resident intermediate, convs_1x.2 from the ReLU tile, both results rounded to fp16 and added
//            exactly as the separate launches round and add them (K5's EPI_ADD on K9's output) -> stored.
//   weights: the fragment streams of K5 v5 (s2m2_pack_frag S2M2_PACK_CONV_FRAG) and K9 (S2M2_PACK_ROWS) as they are, read from global memory
//            straight into MFMA operand registers through the same 8-deep untracked ring with counted waits as conv_frag_kernel.
// Accumulation order = K5 v5's (chunk, tap, channel) and K9's (channel): bit-identical to the three launches (tests/test_hip_convblock.py).
// fp16, C = 128 (PH = 2 / 4) and C = 256 (PH = 2).
#include "common.h"
#include "plan.h"
#include "epilogue.h"
#include <stdlib.h>

namespace s2m2 {

struct CbArgs {
    const half_t* x; half_t* out;
    long long xs, os;                           // elements between pixels
    int N, H, W, tiles_x, tiles_y;
    const raw16_t* w1; const raw16_t* w2;       // convs.0 / convs.2: K5 v5 fragment streams
    const raw16_t* wa; const raw16_t* wb;       // convs_1x.0 / convs_1x.2: K9 fragment order
    const float* b1; const float* b2; const float* ba; const float* bb;   // biases (or zeros)
    const void* zero;
};

template <int C_, int PH_>
struct CbCfg {
    static constexpr int C = C_, PH = PH_, PW = 32, NW = C_ / 32, NT = 64 * NW, KS = 8, CH = 128, NCHUNK = C_ / 128;
    static constexpr int IH = PH + 2, IW = PW + 2, NI = IH * IW, MT1 = (NI + 31) / 32;      // intermediate grid (convs.0 output)
    static constexpr int HH = PH + 4, HW = PW + 4, NHALO = HH * HW;                          // input tile
    static constexpr int RS = CH + 8, TRS = C + 8;                                           // LDS row strides (elements)
    static constexpr int PPX = CH / 8, RPI = NT / PPX, A_IT = (NHALO + RPI - 1) / RPI, AROWS = A_IT * RPI;
    static constexpr size_t A_BYTES = (size_t)AROWS * RS * 2;                                // input tile (one 128-channel chunk)
    static constexpr size_t R_BYTES = (size_t)PH * 32 * TRS * 2;                             // ReLU tile of the 1x1 branch (aliases the input tile)
    static constexpr size_t T_BYTES = (size_t)MT1 * 32 * TRS * 2;                            // intermediate
    static constexpr size_t A_REGION = A_BYTES > R_BYTES ? A_BYTES : R_BYTES;
    static constexpr size_t OFF_T = (A_REGION + 15) / 16 * 16, OFF_B = OFF_T + T_BYTES, LDS_BYTES = OFF_B + 4 * C * 4;
    static_assert(LDS_BYTES <= 160 * 1024 && NT % PPX == 0, "conv block tile");
};

// KS k16 steps of one tap / one chunk of a 1x1 layer: MT pixel tiles from LDS (pixel fragments double buffered) against the ring's fragments.
// `g` = index of the fragment consumed next in this wave's stream of `nfrag` fragments at wf (16-byte units, stride 64 between fragments).
template <int MT, int KS>
__device__ __forceinline__ void cb_steps(float16_t (&acc)[MT], const half_t* a, const int (&poff)[MT], raw16_t (&ring)[KS], const raw16_t* wf,
                                         int& g, int nfrag) {
    Frag<half_t> xf[2][MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) load_frag(xf[0][i], a + poff[i]);
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        if (kk + 1 < KS) {
#pragma unroll
            for (int i = 0; i < MT; ++i) load_frag(xf[(kk + 1) & 1][i], a + poff[i] + (kk + 1) * 16);
        }
        wait_vmcnt<KS - 2>();
        settle(ring[kk]);
        Frag<half_t> wfr;
        wfr.v = __builtin_bit_cast(half8_t, ring[kk]);
#pragma unroll
        for (int i = 0; i < MT; ++i) mma32(acc[i], wfr, xf[kk & 1][i]);
        {
            const int f = g + KS - 1;
            global_load16_async(ring[(kk + KS - 1) % KS], wf + (size_t)(f < nfrag ? f : nfrag - 1) * 64);
        }
        ++g;
    }
}

template <typename CFG>
__global__ __launch_bounds__(CFG::NT) void conv_block_kernel(CbArgs p) {
    constexpr int C = CFG::C, PH = CFG::PH, KS = CFG::KS, RS = CFG::RS, TRS = CFG::TRS, HW = CFG::HW, IW = CFG::IW, MT1 = CFG::MT1, NCHUNK = CFG::NCHUNK;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* Ah = reinterpret_cast<half_t*>(smem);                // input tile [AROWS][RS]; later the ReLU tile [PH * 32][TRS]
    half_t* Rt = reinterpret_cast<half_t*>(smem);
    half_t* Tt = reinterpret_cast<half_t*>(smem + CFG::OFF_T);   // intermediate [MT1 * 32][TRS], row q = iy * 34 + ix
    float* bvec = reinterpret_cast<float*>(smem + CFG::OFF_B);   // b1 | b2 | ba | bb
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // cout tile of this wave
    int bx = blockIdx.x;
    const int tx = bx % p.tiles_x; bx /= p.tiles_x;
    const int ty = bx % p.tiles_y;
    const int n = bx / p.tiles_y;
    const int y0 = ty * PH, x0 = tx * 32;

    for (int i = tid; i < 4 * C; i += CFG::NT) {
        const int k = i / C, c = i - k * C;
        const float* src = k == 0 ? p.b1 : k == 1 ? p.b2 : k == 2 ? p.ba : p.bb;
        bvec[i] = src ? src[c] : 0.f;
    }
    // ---- input tile loader: piece pc of halo pixels prow + RPI * it
    const int pc = tid % CFG::PPX, prow = tid / CFG::PPX;
    const half_t* zp = static_cast<const half_t*>(p.zero);
    auto load_halo = [&](int chunk) __attribute__((always_inline)) {
        raw16_t ra[CFG::A_IT];
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const int hp = prow + CFG::RPI * it;
            const int hy = hp / HW, hx = hp - hy * HW;
            const int yy = y0 - 2 + hy, xx = x0 - 2 + hx;
            const bool ok = hp < CFG::NHALO && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
            const half_t* src = ok ? p.x + ((long long)(n * p.H + yy) * p.W + xx) * p.xs + chunk * CFG::CH + pc * 8 : zp;
            ra[it] = global_load16(src);
        }
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) *reinterpret_cast<raw16_t*>(Ah + (size_t)(prow + CFG::RPI * it) * RS + pc * 8) = ra[it];
    };

    // ---- phase 1: convs.0 on the (PH + 2) x 34 intermediate pixels (+ convs_1x.0 on the patch's own pixels)
    const int nfrag1 = NCHUNK * 9 * KS;
    const raw16_t* wf1 = p.w1 + (size_t)wv * nfrag1 * 64 + lane;
    const raw16_t* wfa = p.wa + (size_t)wv * (C / 16) * 64 + lane;          // K9 order: [cout tile][k16 step][lane]
    raw16_t ring[KS];
    float16_t acc1[MT1], accA[PH];
#pragma unroll
    for (int i = 0; i < MT1; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
#pragma unroll
    for (int i = 0; i < PH; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) accA[i][r] = 0.f;
    int poff1[MT1], poffA[PH];
#pragma unroll
    for (int i = 0; i < MT1; ++i) {
        int q = 32 * i + l31;
        q = q < CFG::NI ? q : CFG::NI - 1;
        const int iy = q / IW, ix = q - iy * IW;
        poff1[i] = (iy * HW + ix) * RS + hi * 8;
    }
#pragma unroll
    for (int i = 0; i < PH; ++i) poffA[i] = ((i + 2) * HW + l31 + 2) * RS + hi * 8;
    // slots 0 .. KS-2 only: slot KS-1 gets its first request from step 0 (a request whose value is never consumed leaves its destination
    // registers free for the allocator while the load is in flight -- the hazard common.h describes)
#pragma unroll
    for (int s = 0; s < KS - 1; ++s) global_load16_async(ring[s], wf1 + (size_t)s * 64);
    int g = 0;
#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
        if (chunk > 0) {
            wait_vmcnt<0>();                                     // the ring's requests land before tracked loads are mixed in
            __syncthreads();                                     // every wave is done with the previous chunk's tile
        }
        load_halo(chunk);
        __syncthreads();
        int ky = 0, kx = 0;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            cb_steps<MT1, KS>(acc1, Ah + (ky * HW + kx) * RS, poff1, ring, wf1, g, nfrag1);
            if (++kx == 3) { kx = 0; ++ky; }
        }
        // convs_1x.0 on this chunk's channels: its 8 fragments as ordinary loads (L2), the patch's own pixels of the tile
        wait_vmcnt<0>();
        {
            raw16_t wr[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) wr[s] = global_load16(wfa + (size_t)(chunk * KS + s) * 64);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Frag<half_t> wfr;
                wfr.v = __builtin_bit_cast(half8_t, wr[s]);
#pragma unroll
                for (int i = 0; i < PH; ++i) {
                    Frag<half_t> xf;
                    load_frag(xf, Ah + poffA[i] + s * 16);
                    mma32(accA[i], wfr, xf);
                }
            }
        }
    }
    // the ring now holds re-requests of the stream's tail (cb_steps clamps): drain, then start convs.2's stream under the epilogues
    wait_vmcnt<0>();
#pragma unroll
    for (int s = 0; s < KS; ++s) settle(ring[s]);
    const int nfrag2 = NCHUNK * 9 * KS;
    const raw16_t* wf2 = p.w2 + (size_t)wv * nfrag2 * 64 + lane;
    __syncthreads();                                             // every wave is done with the input tile (the ReLU tile aliases it)
    // ---- epilogue 1: bias + GELU -> intermediate (zero outside the image / past the grid); ReLU(convs_1x.0) -> ReLU tile
#pragma unroll
    for (int i = 0; i < MT1; ++i) {
        const int q = 32 * i + l31;
        const int iy = q / IW, ix = q - iy * IW;
        const int yy = y0 - 1 + iy, xx = x0 - 1 + ix;
        const bool inside = q < CFG::NI && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        half_t* trow = Tt + (size_t)q * TRS + wv * 32 + 4 * hi;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4_t bv = *reinterpret_cast<const float4_t*>(bvec + wv * 32 + 8 * gq + 4 * hi);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc1[i][4 * gq + e] + bv[e];
            const float2_t r0 = fast_gelu16x2((float2_t){v[0], v[1]}), r1 = fast_gelu16x2((float2_t){v[2], v[3]});
            half4_t h = {from_f32<half_t>(r0.x * 1.0f), from_f32<half_t>(r0.y * 1.0f), from_f32<half_t>(r1.x * 1.0f), from_f32<half_t>(r1.y * 1.0f)};
            if (!inside) h = half4_t{(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
            *reinterpret_cast<half4_t*>(trow + 8 * gq) = h;
        }
    }
#pragma unroll
    for (int i = 0; i < PH; ++i) {
        half_t* rrow = Rt + (size_t)(32 * i + l31) * TRS + wv * 32 + 4 * hi;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4_t bv = *reinterpret_cast<const float4_t*>(bvec + 2 * C + wv * 32 + 8 * gq + 4 * hi);
            half4_t h;
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = from_f32<half_t>(fmaxf(accA[i][4 * gq + e] + bv[e], 0.f) * 1.0f);
            *reinterpret_cast<half4_t*>(rrow + 8 * gq) = h;
        }
    }
#pragma unroll
    for (int s = 0; s < KS - 1; ++s) global_load16_async(ring[s], wf2 + (size_t)s * 64);
    __syncthreads();
    // ---- phase 2: convs.2 on the patch from the intermediate; convs_1x.2 from the ReLU tile
    float16_t acc2[PH], accB[PH];
#pragma unroll
    for (int i = 0; i < PH; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc2[i][r] = 0.f; accB[i][r] = 0.f; }
    int poff2[PH];
#pragma unroll
    for (int i = 0; i < PH; ++i) poff2[i] = (i * IW + l31) * TRS + hi * 8;
    g = 0;
#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
        int ky = 0, kx = 0;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            cb_steps<PH, KS>(acc2, Tt + (ky * IW + kx) * TRS + chunk * CFG::CH, poff2, ring, wf2, g, nfrag2);
            if (++kx == 3) { kx = 0; ++ky; }
        }
    }
    wait_vmcnt<0>();
#pragma unroll
    for (int s = 0; s < KS; ++s) settle(ring[s]);
    {
        const raw16_t* wfb = p.wb + (size_t)wv * (C / 16) * 64 + lane;
#pragma unroll 1
        for (int c8 = 0; c8 < C / 16; c8 += KS) {
            raw16_t wr[KS];
#pragma unroll
            for (int s = 0; s < KS; ++s) wr[s] = global_load16(wfb + (size_t)(c8 + s) * 64);
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                Frag<half_t> wfr;
                wfr.v = __builtin_bit_cast(half8_t, wr[s]);
#pragma unroll
                for (int i = 0; i < PH; ++i) {
                    Frag<half_t> xf;
                    load_frag(xf, Rt + (size_t)(32 * i + l31) * TRS + hi * 8 + (c8 + s) * 16);
                    mma32(accB[i], wfr, xf);
                }
            }
        }
    }
    // ---- out = fp16(fp16(convs.2 + b2) + fp16(convs_1x.2 + bb)): K5's EPI_ADD on K9's output
#pragma unroll
    for (int i = 0; i < PH; ++i) {
        const int yy = y0 + i, xx = x0 + l31;
        if (yy < p.H && xx < p.W) {
            half_t* orow = p.out + ((long long)(n * p.H + yy) * p.W + xx) * p.os + wv * 32 + 4 * hi;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4_t b2 = *reinterpret_cast<const float4_t*>(bvec + C + wv * 32 + 8 * gq + 4 * hi);
                const float4_t bb = *reinterpret_cast<const float4_t*>(bvec + 3 * C + wv * 32 + 8 * gq + 4 * hi);
                half4_t h;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const half_t m = from_f32<half_t>((acc2[i][4 * gq + e] + b2[e]) * 1.0f);
                    const half_t s1 = from_f32<half_t>((accB[i][4 * gq + e] + bb[e]) * 1.0f);
                    h[e] = from_f32<half_t>((float)m + (float)s1);
                }
                *reinterpret_cast<half4_t*>(orow + 8 * gq) = h;
            }
        }
    }
}

template <int C, int PH>
static int launch_cb(const CbArgs& a0, hipStream_t st) {
    using CFG = CbCfg<C, PH>;
    CbArgs a = a0;
    a.tiles_x = (a.W + 31) / 32;
    a.tiles_y = (a.H + PH - 1) / PH;
    auto kern = conv_block_kernel<CFG>;
    static size_t granted[kMaxDevices] = {};
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, granted, "conv_block")) return 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.N * a.tiles_x * a.tiles_y)), dim3(CFG::NT), CFG::LDS_BYTES, st, a);
    return check_launch("conv_block");
}

}  // namespace s2m2

extern "C" int s2m2_conv_block_supported(int C, int H, int W, int dtype) {
    return dtype == S2M2_F16 && (C == 128 || C == 256) && H >= 1 && W >= 1 && (long long)H * W <= 160 * 192;
}

static int conv_block_impl(const s2m2_convblock_desc* d, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(d, "conv_block: null descriptor");
    S2M2_REQUIRE(d->dtype == S2M2_F16 && (d->C == 128 || d->C == 256), "conv_block: C=%d dtype=%d (fp16, C = 128 / 256)", d->C, d->dtype);
    S2M2_REQUIRE(d->x && d->out && d->x != d->out, "conv_block: x / out must be distinct non-null tensors");
    S2M2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && (long long)d->N * d->H * d->W < (1LL << 24), "conv_block: bad shape");
    S2M2_REQUIRE(d->x_stride >= d->C && d->x_stride % 8 == 0 && d->out_stride >= d->C && d->out_stride % 4 == 0, "conv_block: pixel strides");
    S2M2_REQUIRE(d->w_conv0 && d->w_conv2 && d->w_1x0 && d->w_1x2, "conv_block: null weight");
    CbArgs a;
    a.x = static_cast<const half_t*>(d->x); a.out = static_cast<half_t*>(d->out); a.xs = d->x_stride; a.os = d->out_stride;
    a.N = d->N; a.H = d->H; a.W = d->W;
    a.w1 = static_cast<const raw16_t*>(d->w_conv0); a.w2 = static_cast<const raw16_t*>(d->w_conv2);
    a.wa = static_cast<const raw16_t*>(d->w_1x0); a.wb = static_cast<const raw16_t*>(d->w_1x2);
    a.b1 = d->b_conv0; a.b2 = d->b_conv2; a.ba = d->b_1x0; a.bb = d->b_1x2;
    a.zero = zero_page();
    S2M2_REQUIRE(a.zero, "conv_block: cannot allocate the zero page");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->C == 256) return launch_cb<256, 2>(a, st);
    // C = 128: 4-row patches while they give every CU about a block, 2-row patches on the smaller grids
    const long long b4 = (long long)d->N * ((d->W + 31) / 32) * ((d->H + 3) / 4);
    return (d->patch_rows == 4 || (d->patch_rows != 2 && b4 >= 128)) ? launch_cb<128, 4>(a, st) : launch_cb<128, 2>(a, st);
}
extern "C" int s2m2_conv_block(const s2m2_convblock_desc* d, void* stream) {
    return s2m2::plan_dispatch_desc<s2m2_convblock_desc>("s2m2_conv_block", &conv_block_impl, d, stream);
}
