// K12 -- spatial convolutions on NARROW inputs (8 or 16 channels: one or two 16-byte pieces per pixel) in the direct style (fp16).
//
// Replaces the K5 v1 launches (conv.hip: dense-K im2col tiles, both operands staged through LDS behind a block barrier per K tile) of
//   UpsampleMask1x   conv_disp.0 | conv_rgb.0 merged: ConvTranspose2d(k 3, s 1) on (disp, rgb) -> 32, ReLU, full resolution   submodules.py:124-129,139-141
//   LocalRefiner     disp_feat.0 | conf_occ_feat.0 merged: Conv2d(k 3) on (disp, conf, occ) -> 96 + 64, GELU, 1/4 resolution   refinenet.py:93-101,141-142
//   CNNEncoder       conv1_down.0: Conv2d(16 -> 64, k 5, s 2), GELU, full -> half resolution                                 submodules.py:69-71
// These layers are memory-bound (8 - 16 input channels against 32 - 160 output channels: 5 / 25 MFMA k16 steps per 32 pixels), the v1 tile
// spent its time re-gathering the im2col operand and in per-K-tile barriers.
//
// One block = 4 waves = an output patch of (4 MT) rows x 32 columns.  The input patch + halo is loaded ONCE into LDS, pixel-major, 16 / 32
// bytes per pixel; wave v owns output rows v MT .. v MT + MT - 1 of the patch (MT 32-pixel MFMA tiles) and ALL output channels.  With 8
// input channels a k16 step is two taps: lane (l % 32, l / 32) reads the 16-byte pixel of tap 2s + l / 32 of ITS output pixel straight out of
// the tile (tap 9 of a 3x3 kernel: a zero slot; the packed weight is zero there as well); with 16 channels a step is one tap, the lane halves
// take the channel halves.  The weight fragments (pack.pw_frag of the K5 K-order-0 matrix: K = (tap, channel), zero-padded to k16 steps and
// 32-cout tiles) go from global memory straight into MFMA operand registers -- they are a few KB, L2-resident, shared by every block.
// Epilogue per wave, no block barrier: bias + activation in registers -> a wave-private staging tile [pixel][cout] -> 16-byte NHWC stores.
//
// Second form, same entry point (conv_px_kernel): 3x3 layers with FEW OUTPUT channels on wider inputs -- the other layers K5 ran on its
// LDS-staged tiles because they are no multiple of the 128-cout blocks of the fragment-stream kernel (K5 v5):
//   UpsampleMask1x   conv_concat.0: Conv2d(cat(32, 16) -> 48, k 3), ReLU, full resolution                                  submodules.py:133-137,143
//   LocalRefiner     disp_feat.2: Conv2d(96 -> 96, k 3)                                                                    refinenet.py:93-96
//                    disp_update.2 | conf_occ_update.2 merged: Conv2d(2C -> 1 + 2, k 3), out_feat of GlobalRefiner (C -> 1)   refinenet.py:61-66,108-118
// The same pixel split (a wave = its own output rows x ALL couts, up to three 32-cout tiles), the input patch + halo in LDS per chunk of
// CH = 48 / 96 / 64 channels (row stride CH + 8 halfs), the K loop of a chunk = 9 taps x CH / 16 steps x NTL fragments through the same
// register ring; further chunks (Cin = 128 / 256: 2 / 4 of 64 channels) re-load the tile behind a block barrier.  Fragment order per cout
// tile: (chunk, tap, k16 step) = the K order 0 of K5 for one chunk, columns permuted chunk-major for several (pack.narrow_frag).
// Fused 1x1 head (UpsampleMask1x: conv_concat.0 -> ReLU -> conv_concat.2, ConvTranspose2d(48 -> 9, k 1), submodules.py:133-137,143-144): in the
// accumulator layout of D[cout][pixel], lane (pixel l % 32, half l / 32) of cout tile j holds channels 32 j + 8 g + 4 (l / 32) + e, g < 4, e < 4 --
// for g = 2p, 2p + 1 that IS an MFMA pixel fragment of the 1x1 layer, provided that layer's K columns are packed in the same order (pack.head_frag:
// k16 step (j, p), lane half, then (g parity, e)).  So the activated, fp16-rounded outputs of the 3x3 layer feed the head's MFMAs straight from
// registers: the 48-channel tensor is never written, the head is four more MFMAs per 32 pixels (the fourth on the zero padding of cout tile 1).
#include <hip/hip_runtime.h>

#include "common.h"
#include "plan.h"
#include "epilogue.h"

namespace s2m2 {

struct NarrowArgs {
    const void* x;
    long long xstride;                  // pixel stride of x (elements)
    const void* x1;                     // conv_px_kernel: second source (channels [c0, Cin)), or x again
    long long x1stride;
    int c0, Cin, nchunk;                // channels of x; all input channels; channel chunks of CH
    int N, H, W, Ho, Wo;
    const void* w;                      // fragment order: [cout tile][k16 step][lane] x 16 bytes
    const float* bias;
    const void* zero;
    void* out;
    long long out_stride;
    int Cout, act;
    const void* w2;                     // conv_px_kernel, fused 1x1 head: fragments of the (Cout2, Cout) matrix in accumulator order (pack.head_frag)
    const float* bias2;
    int Cout2;                          // 0: no head, out holds the Cout channels of the 3x3 layer; > 0: out holds the Cout2 channels of the head
};

template <int KH_, int KW_, int S_, int CIN_, int MT_, int NTL_, int NWN_ = 1>
struct NarrowCfg {
    static constexpr int KH = KH_, KW = KW_, S = S_, CIN = CIN_, MT = MT_, NTL = NTL_;
    static constexpr int NW = 4, NT = 64 * NW;
    // NWN waves share a set of MT output rows and split the couts of a group (NWN * NTL tiles) between them: a wave's weight stream is then
    // 1 / NWN of the layer's and feeds NWN times the pixel tiles (5x5 layer, 25 KB of fragments per cout tile: 67.1 -> 65.1 us)
    static constexpr int NWN = NWN_, NWM = NW / NWN_;
    static constexpr int PW = 32, PH = NWM * MT;                          // output patch
    static constexpr int HW = (PW - 1) * S + KW, HH = (PH - 1) * S + KH;  // input patch + halo
    // stride 2: the columns of a tile row are stored even ones first, then the odd ones -- the 32 lanes of a fragment read (columns 2 l + kx)
    // then touch CONSECUTIVE slots instead of every second one (which is an 8-way LDS bank conflict on 16-byte reads)
    static constexpr int HWE = (HW + 1) / 2;                              // even columns of a tile row
    static __device__ __forceinline__ int col_slot(int hx) { return S == 2 ? (hx & 1) * HWE + (hx >> 1) : hx; }
    static constexpr int PPX = CIN / 8;                                   // 16-byte pieces per input pixel
    static constexpr int NPIECE = HW * HH * PPX;
    static constexpr int A_IT = (NPIECE + NT - 1) / NT;
    static constexpr int NTAP = KH * KW, K = NTAP * CIN, NS = (K + 15) / 16;
    static constexpr bool RESIDENT = NS * MT <= 12;                       // the pixel fragments of a wave stay in registers across cout groups
    static constexpr int SM = MT < 2 ? MT : 2;                            // pixel tiles a wave stages at a time
    struct Stage {                                                        // (names used by stage_tile / CoutRegs)
        static constexpr int MT = SM, NTL = NTL_, WM = 32 * SM, WN = 32 * NTL_, CRS = WN + 8;
    };
    static constexpr int WN = Stage::WN, CRS = Stage::CRS;                // couts of a wave per group; staging row stride
    static constexpr int CPR = WN / 8, C_IT = Stage::WM * CPR / 64;       // 16-byte pieces per staged row / per lane and staging pass
    static constexpr size_t A_BYTES = (size_t)(NPIECE + 1) * 16;          // + one zero slot
    static constexpr size_t STG_BYTES = (size_t)NW * Stage::WM * CRS * sizeof(half_t);
    static constexpr size_t LDS_BYTES = A_BYTES + STG_BYTES;
    static_assert((CIN == 8 || CIN == 16) && (Stage::WM * CPR) % 64 == 0 && MT % SM == 0 && NW % NWN == 0 && LDS_BYTES <= 80 * 1024,
                  "unsupported narrow-input tile");
};

template <typename CFG>
__global__ __launch_bounds__(CFG::NT) void conv_narrow_kernel(NarrowArgs p, int tiles_x, int tiles_y, int gpb) {
    using T = half_t;
    constexpr int KW = CFG::KW, S = CFG::S, MT = CFG::MT, NTL = CFG::NTL, NS = CFG::NS, HW = CFG::HW, CRS = CFG::CRS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    raw16_t* A = reinterpret_cast<raw16_t*>(smem);                  // [HH][HW][PPX] pieces, then one zero slot
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pm = wv / CFG::NWN, cn = wv - pm * CFG::NWN;          // row set / cout share of this wave
    T* stg = reinterpret_cast<T*>(smem + CFG::A_BYTES) + (size_t)wv * CFG::Stage::WM * CRS;     // wave-private staging tile [32 SM][CRS]
    int bx = blockIdx.x;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int n = bx / tiles_y;
    const int oy0 = ty * CFG::PH, ox0 = tx * CFG::PW;
    const int iy0 = oy0 * S - CFG::KH / 2, ix0 = ox0 * S - KW / 2;

    // ---- 1. input patch + halo -> LDS (zero outside the image: the convolution's padding)
    const T* xin = static_cast<const T*>(p.x);
    raw16_t ra[CFG::A_IT];
#pragma unroll
    for (int it = 0; it < CFG::A_IT; ++it) {
        const int idx = tid + CFG::NT * it, pix = idx / CFG::PPX, part = idx - pix * CFG::PPX;
        const int hy = pix / HW, hx = pix - hy * HW;
        const int yy = iy0 + hy, xx = ix0 + hx;
        const bool ok = idx < CFG::NPIECE && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
        ra[it] = global_load16(ok ? xin + (((long long)n * p.H + yy) * p.W + xx) * p.xstride + part * 8 : static_cast<const T*>(p.zero));
    }
#pragma unroll
    for (int it = 0; it < CFG::A_IT; ++it) {
        const int idx = tid + CFG::NT * it;
        if (idx < CFG::NPIECE) {
            const int pix = idx / CFG::PPX, part = idx - pix * CFG::PPX;
            const int hy = pix / HW, hx = pix - hy * HW;
            A[(hy * HW + CFG::col_slot(hx)) * CFG::PPX + part] = ra[it];
        }
    }
    if (tid == 0) A[CFG::NPIECE] = (raw16_t){0.f, 0.f, 0.f, 0.f};
    __syncthreads();

    // ---- 2. pixel fragment of MFMA tile i (output row pm * MT + i of the patch), k16 step s: one 16-byte LDS read per lane
    auto xfrag = [&](int i, int s) __attribute__((always_inline)) {
        const int r = (pm * MT + i) * S;
        int piece;
        if constexpr (CFG::CIN == 8) {                              // two taps per step, the lane half picks the tap
            const int t0 = 2 * s, t1 = 2 * s + 1;
            const int o0 = ((r + t0 / KW) * HW + CFG::col_slot(l31 * S + t0 % KW));
            const int o1 = t1 < CFG::NTAP ? ((r + t1 / KW) * HW + CFG::col_slot(l31 * S + t1 % KW)) : CFG::NPIECE;
            piece = hi ? o1 : o0;
        } else {                                                    // one tap per step, the lane half picks the channel half
            piece = ((r + s / KW) * HW + CFG::col_slot(l31 * S + s % KW)) * 2 + hi;
        }
        Frag<T> f;
        f.v = __builtin_bit_cast(half8_t, A[piece]);
        return f;
    };
    Frag<T> xres[CFG::RESIDENT ? MT : 1][CFG::RESIDENT ? NS : 1];
    if constexpr (CFG::RESIDENT) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int s = 0; s < NS; ++s) xres[i][s] = xfrag(i, s);
    }

    // ---- 3. groups of NWN * NTL 32-cout tiles: K loop (NS steps x MT tiles x NTL couts, no synchronisation), then the wave's own epilogue
    // (small grids: the groups are spread over blockIdx.y -- every block then re-loads the few KB of its input tile and takes gpb of them)
    const int ngroups = (p.Cout + CFG::NWN * CFG::WN - 1) / (CFG::NWN * CFG::WN);
    const int cg0 = blockIdx.y * gpb, cg1 = cg0 + gpb < ngroups ? cg0 + gpb : ngroups;
    T* outp = static_cast<T*>(p.out);
    for (int cg = cg0; cg < cg1; ++cg) {
        const int cout0 = (cg * CFG::NWN + cn) * CFG::WN;            // first cout of this wave in this group
        const raw16_t* wq = static_cast<const raw16_t*>(p.w) + (size_t)(cout0 / 32) * NS * 64 + lane;
        CoutRegs<typename CFG::Stage> bias;
        bias.load(p.bias, p.zero, p.Cout, cout0, 0, lane);
        float16_t acc[MT][NTL];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NTL; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // weight fragments in consumption order f = (step, cout tile): a register ring of D untracked loads with counted waits (common.h) --
        // left to the compiler, two or three requests were in flight and every k16 step waited for an L2 round trip.  D - 1 requests up
        // front, then every fragment issues the request D - 1 ahead of it (none past the end: the counted waits of the fully unrolled tail
        // shrink instead).  Every request is consumed by an MFMA before its slot is requested again: a request whose result is overwritten
        // unread -- or only "used" by a final settle() -- is a dead definition to the register allocator, which then reuses or copies its
        // destination while the load is still landing (seen as a memory fault in the first version, which requested fragment D - 1 twice;
        // tools/check_isa.py flags the copies).
        constexpr int NF = NS * NTL, D = NF < 8 ? NF : 8;
        static_assert(D >= 2, "ring depth");
        raw16_t ring[D];
        auto fptr = [&](int f) __attribute__((always_inline)) { return wq + (size_t)((f % NTL) * NS + f / NTL) * 64; };
#pragma unroll
        for (int f = 0; f < D - 1; ++f) global_load16_async(ring[f], fptr(f));
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            Frag<T> xf[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if constexpr (CFG::RESIDENT) xf[i] = xres[i][s]; else xf[i] = xfrag(i, s);
            }
#pragma unroll
            for (int j = 0; j < NTL; ++j) {
                const int f = s * NTL + j;
                if (f + D - 1 < NF) global_load16_async(ring[(f + D - 1) % D], fptr(f + D - 1));   // the slot consumed one fragment ago (fragment 0: the last free one)
                wait_vmcnt_n(NF - 1 - f < D - 1 ? NF - 1 - f : D - 1);   // requests younger than fragment f's (f is a constant after unrolling)
                settle(ring[f % D]);
                Frag<T> wf;
                wf.v = __builtin_bit_cast(half8_t, ring[f % D]);
#pragma unroll
                for (int i = 0; i < MT; ++i) mma32(acc[i][j], wf, xf[i]);
            }
        }
        // the staging tile is private to the wave: LDS operations of one wave execute in order, no block barrier -- only the compiler must
        // not move the reads above the writes (or the next pass's writes above these reads)
        using STG = typename CFG::Stage;
#pragma unroll
        for (int h = 0; h < MT / CFG::SM; ++h) {
            const float16_t (&sub)[CFG::SM][NTL] = *reinterpret_cast<const float16_t (*)[CFG::SM][NTL]>(&acc[h * CFG::SM]);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            switch (p.act) {                                        // block-uniform
                case S2M2_ACT_GELU: stage_tile<STG, T, S2M2_ACT_GELU>(sub, stg, bias, 1.0f, 0, 0, lane); break;
                case S2M2_ACT_RELU: stage_tile<STG, T, S2M2_ACT_RELU>(sub, stg, bias, 1.0f, 0, 0, lane); break;
                default: stage_tile<STG, T, S2M2_ACT_NONE>(sub, stg, bias, 1.0f, 0, 0, lane); break;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int it = 0; it < CFG::C_IT; ++it) {
                const int q = lane + 64 * it, row = q / CFG::CPR, pc = q - row * CFG::CPR;
                const int i = h * CFG::SM + (row >> 5), px = row & 31;
                const int oy = oy0 + pm * MT + i, ox = ox0 + px, co = cout0 + pc * 8;
                if (oy < p.Ho && ox < p.Wo && co < p.Cout) {
                    const raw16_t v = *reinterpret_cast<const raw16_t*>(stg + (size_t)row * CRS + pc * 8);
                    *reinterpret_cast<raw16_t*>(outp + (((long long)n * p.Ho + oy) * p.Wo + ox) * p.out_stride + co) = v;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// conv_px_kernel: 3x3 stride 1, Cin = nchunk * CH (CH = 48 / 96 / 64), NTL cout tiles per block (blockIdx.y: which), one or two sources
template <int CH_, int NTL_, int MT_, int NW_ = 4>
struct PxCfg {
    static constexpr int CH = CH_, NTL = NTL_, MT = MT_, KS = CH_ / 16, NTAP = 9;
    // NW: waves per block = output rows per block / MT.  The layers at 1/4 resolution are chains of memory latencies (a tile load per chunk,
    // then a K loop of a few hundred cycles): FEWER waves per block = more, smaller blocks per CU whose chains interleave
    static constexpr int NW = NW_, NT = 64 * NW;
    static constexpr int PW = 32, PH = NW * MT, HW = PW + 2, HH = PH + 2;
    static constexpr int RS = CH + 8;                                     // LDS row stride (elements): 16 bytes of padding per halo pixel
    static constexpr int PPX = CH / 8, NPIECE = HW * HH * PPX, A_IT = (NPIECE + NT - 1) / NT;
    static constexpr int NF = NTAP * KS * NTL;                            // fragments of a chunk, per wave
    static constexpr int D = 8;                                           // ring depth
    static constexpr int SM = MT < 2 ? MT : 2;
    struct Stage {
        static constexpr int MT = SM, NTL = NTL_, WM = 32 * SM, WN = 32 * NTL_, CRS = WN + 8;
    };
    struct Stage2 {                                                       // the fused 1x1 head's tile: 32 couts per pixel
        static constexpr int MT = SM, NTL = 1, WM = 32 * SM, WN = 32, CRS = 40;
    };
    static constexpr int WN = Stage::WN, CRS = Stage::CRS, CPR = WN / 8, C_IT = Stage::WM * CPR / 64;
    static constexpr size_t A_BYTES = (size_t)HW * HH * RS * sizeof(half_t);
    static constexpr size_t STG_BYTES = (size_t)NW * Stage::WM * CRS * sizeof(half_t);
    static constexpr size_t LDS_BYTES = A_BYTES + STG_BYTES;
    static_assert(CH % 16 == 0 && (Stage::WM * CPR) % 64 == 0 && MT % SM == 0 && LDS_BYTES <= 80 * 1024, "unsupported pixel-split tile");
};

template <typename CFG, bool HEAD = false>
__global__ __launch_bounds__(CFG::NT) void conv_px_kernel(NarrowArgs p, int tiles_x, int tiles_y) {
    using T = half_t;
    constexpr int MT = CFG::MT, NTL = CFG::NTL, KS = CFG::KS, HW = CFG::HW, RS = CFG::RS, CRS = CFG::CRS, D = CFG::D, NF = CFG::NF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* A = reinterpret_cast<T*>(smem);                              // [HH][HW][RS] halo tile of one channel chunk
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    T* stg = reinterpret_cast<T*>(smem + CFG::A_BYTES) + (size_t)wv * CFG::Stage::WM * CRS;
    int bx = blockIdx.x;
    const int tx = bx % tiles_x; bx /= tiles_x;
    const int ty = bx % tiles_y;
    const int n = bx / tiles_y;
    const int oy0 = ty * CFG::PH, ox0 = tx * CFG::PW;

    // blockIdx.y: which group of NTL cout tiles (small grids: the cout tiles of a layer are spread over blocks that each re-load the
    // input tile -- three times the blocks, each a third as long, instead of a grid that ends in a quarter-full round)
    const int cout0 = blockIdx.y * CFG::WN;
    CoutRegs<typename CFG::Stage> bias;
    bias.load(p.bias, p.zero, p.Cout, cout0, 0, lane);
    float16_t acc[MT][NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NTL; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nstot = p.nchunk * CFG::NTAP * KS;                    // k16 steps of a cout tile
    const raw16_t* wbase = static_cast<const raw16_t*>(p.w) + (size_t)(cout0 / 32) * nstot * 64 + lane;
    const T* x0 = static_cast<const T*>(p.x);
    const T* x1 = static_cast<const T*>(p.x1);

    for (int chunk = 0; chunk < p.nchunk; ++chunk) {
        // ---- halo tile of this chunk (zero outside the image / beyond Cin)
        if (chunk > 0) __syncthreads();                             // every wave is done with the previous chunk's tile
        raw16_t ra[CFG::A_IT];
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const int idx = tid + CFG::NT * it, pix = idx / CFG::PPX, part = idx - pix * CFG::PPX;
            const int hy = pix / HW, hx = pix - hy * HW;
            const int yy = oy0 - 1 + hy, xx = ox0 - 1 + hx;
            const int c = chunk * CFG::CH + part * 8;
            const bool ok = idx < CFG::NPIECE && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W && c < p.Cin;
            const long long pixel = ((long long)n * p.H + yy) * p.W + xx;
            const T* src = c < p.c0 ? x0 + pixel * p.xstride + c : x1 + pixel * p.x1stride + (c - p.c0);
            ra[it] = global_load16(ok ? src : static_cast<const T*>(p.zero));
        }
#pragma unroll
        for (int it = 0; it < CFG::A_IT; ++it) {
            const int idx = tid + CFG::NT * it, pix = idx / CFG::PPX, part = idx - pix * CFG::PPX;
            if (idx < CFG::NPIECE) *reinterpret_cast<raw16_t*>(A + (size_t)pix * RS + part * 8) = ra[it];
        }
        __syncthreads();

        // ---- K loop of the chunk: fragments in consumption order f = (tap, k16 step, cout tile) through the register ring (see above)
        const raw16_t* wq = wbase + (size_t)chunk * CFG::NTAP * KS * 64;
        raw16_t ring[D];
        auto fptr = [&](int f) __attribute__((always_inline)) { return wq + ((size_t)(f % NTL) * nstot + f / NTL) * 64; };
#pragma unroll
        for (int f = 0; f < D - 1; ++f) global_load16_async(ring[f], fptr(f));
        const T* arow = A + (size_t)(wv * MT * HW + l31) * RS + hi * 8;
#pragma unroll
        for (int tap = 0; tap < CFG::NTAP; ++tap) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag<T> xf[MT];
#pragma unroll
                for (int i = 0; i < MT; ++i) load_frag(xf[i], arow + (size_t)((i + tap / 3) * HW + tap % 3) * RS + ks * 16);
#pragma unroll
                for (int j = 0; j < NTL; ++j) {
                    const int f = (tap * KS + ks) * NTL + j;
                    if (f + D - 1 < NF) global_load16_async(ring[(f + D - 1) % D], fptr(f + D - 1));
                    wait_vmcnt_n(NF - 1 - f < D - 1 ? NF - 1 - f : D - 1);
                    settle(ring[f % D]);
                    Frag<T> wf;
                    wf.v = __builtin_bit_cast(half8_t, ring[f % D]);
#pragma unroll
                    for (int i = 0; i < MT; ++i) mma32(acc[i][j], wf, xf[i]);
                }
            }
        }
    }

    // ---- epilogue per wave (wave-private staging tile, as above)
    using STG = typename CFG::Stage;
    T* outp = static_cast<T*>(p.out);
    if constexpr (HEAD) {
        // fused 1x1 head: activation of the 3x3 layer in registers, rounded to fp16 where the unfused pair of launches rounds it (its store),
        // then D2[cout2][pixel] += W2 fragment (j, p) . {quads 2p, 2p + 1 of cout tile j}: the accumulator layout is the operand layout
        using Stage2 = typename CFG::Stage2;
        constexpr int KS2 = NTL * 2;                                // k16 steps of the head (zero weights beyond Cout)
        const raw16_t* w2q = static_cast<const raw16_t*>(p.w2) + lane;
        Frag<T> w2f[KS2];
#pragma unroll
        for (int s2 = 0; s2 < KS2; ++s2) w2f[s2].v = __builtin_bit_cast(half8_t, global_load16(w2q + s2 * 64));
        CoutRegs<Stage2> bias2;
        bias2.load(p.bias2, p.zero, p.Cout2, 0, 0, lane);
        const bool relu = p.act == S2M2_ACT_RELU;
#pragma unroll
        for (int h = 0; h < MT / CFG::SM; ++h) {
            float16_t acc2[CFG::SM][1];
#pragma unroll
            for (int i = 0; i < CFG::SM; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[i][0][r] = 0.f;
#pragma unroll
                for (int j = 0; j < NTL; ++j)
#pragma unroll
                    for (int pp = 0; pp < 2; ++pp) {
                        Frag<T> yf;
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int g = 2 * pp + q;
                            const raw16_t bv = bias.v[j][g];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float a = acc[h * CFG::SM + i][j][4 * g + e] + bv[e];
                                yf.v[4 * q + e] = from_f32<half_t>(relu ? fmaxf(a, 0.f) : a);       // (NONE / RELU only: checked by the host entry)
                            }
                        }
                        mma32(acc2[i][0], w2f[2 * j + pp], yf);
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            stage_tile<Stage2, T, S2M2_ACT_NONE>(acc2, stg, bias2, 1.0f, 0, 0, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            constexpr int CPR2 = 4, C_IT2 = Stage2::WM * CPR2 / 64;  // 32 staged couts = 4 pieces per pixel
#pragma unroll
            for (int it = 0; it < C_IT2; ++it) {
                const int q = lane + 64 * it, row = q / CPR2, pc = q - row * CPR2;
                const int i = h * CFG::SM + (row >> 5), px = row & 31;
                const int oy = oy0 + wv * MT + i, ox = ox0 + px, co = pc * 8;
                if (oy < p.Ho && ox < p.Wo && co < p.Cout2) {
                    const raw16_t v = *reinterpret_cast<const raw16_t*>(stg + (size_t)row * Stage2::CRS + pc * 8);
                    *reinterpret_cast<raw16_t*>(outp + (((long long)n * p.Ho + oy) * p.Wo + ox) * p.out_stride + co) = v;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int h = 0; h < MT / CFG::SM; ++h) {
        const float16_t (&sub)[CFG::SM][NTL] = *reinterpret_cast<const float16_t (*)[CFG::SM][NTL]>(&acc[h * CFG::SM]);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        switch (p.act) {                                            // block-uniform
            case S2M2_ACT_GELU: stage_tile<STG, T, S2M2_ACT_GELU>(sub, stg, bias, 1.0f, 0, 0, lane); break;
            case S2M2_ACT_RELU: stage_tile<STG, T, S2M2_ACT_RELU>(sub, stg, bias, 1.0f, 0, 0, lane); break;
            default: stage_tile<STG, T, S2M2_ACT_NONE>(sub, stg, bias, 1.0f, 0, 0, lane); break;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int it = 0; it < CFG::C_IT; ++it) {
            const int q = lane + 64 * it, row = q / CFG::CPR, pc = q - row * CFG::CPR;
            const int i = h * CFG::SM + (row >> 5), px = row & 31;
            const int oy = oy0 + wv * MT + i, ox = ox0 + px, co = cout0 + pc * 8;
            if (oy < p.Ho && ox < p.Wo && co < p.Cout) {
                const raw16_t v = *reinterpret_cast<const raw16_t*>(stg + (size_t)row * CRS + pc * 8);
                *reinterpret_cast<raw16_t*>(outp + (((long long)n * p.Ho + oy) * p.Wo + ox) * p.out_stride + co) = v;
            }
        }
    }
}

template <int CH, int NTL, int MT, int NW = 4, bool HEAD = false>
static int launch_px(const NarrowArgs& a, hipStream_t st) {
    using CFG = PxCfg<CH, NTL, MT, NW>;
    auto kern = conv_px_kernel<CFG, HEAD>;
    static size_t lds_granted[kMaxDevices] = {};
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted, "conv_narrow")) return 1;
    const int tx = (a.Wo + CFG::PW - 1) / CFG::PW, ty = (a.Ho + CFG::PH - 1) / CFG::PH;
    const long long nblk = (long long)a.N * tx * ty;
    if (nblk >= (1LL << 31)) return set_error("conv_narrow: %lld blocks", nblk);
    const int gy = (a.Cout + CFG::WN - 1) / CFG::WN;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)gy), dim3(CFG::NT), CFG::LDS_BYTES, st, a, tx, ty);
    return check_launch("conv_narrow");
}

template <int KH, int KW, int S, int CIN, int MT, int NTL, int NWN>
static int launch_narrow(const NarrowArgs& a, hipStream_t st) {
    using CFG = NarrowCfg<KH, KW, S, CIN, MT, NTL, NWN>;
    auto kern = conv_narrow_kernel<CFG>;
    static size_t lds_granted[kMaxDevices] = {};
    if (reserve_lds(reinterpret_cast<const void*>(kern), CFG::LDS_BYTES, lds_granted, "conv_narrow")) return 1;
    const int tx = (a.Wo + CFG::PW - 1) / CFG::PW, ty = (a.Ho + CFG::PH - 1) / CFG::PH;
    const long long nblk = (long long)a.N * tx * ty;
    if (nblk >= (1LL << 31)) return set_error("conv_narrow: %lld blocks", nblk);
    // cout groups per block: all of them on a grid that fills the chip by itself, one where every CU would get less than ~4 blocks
    const int ngroups = (a.Cout + CFG::NWN * CFG::WN - 1) / (CFG::NWN * CFG::WN);
    const int gpb = nblk >= 1024 ? ngroups : 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk, (unsigned)((ngroups + gpb - 1) / gpb)), dim3(CFG::NT), CFG::LDS_BYTES, st, a, tx, ty, gpb);
    return check_launch("conv_narrow");
}

// 8 output rows per block where that still gives every CU several blocks, else 4
static bool narrow_tall(const NarrowArgs& a) {
    static const int force_mt = getenv("S2M2_NARROW_MT") ? atoi(getenv("S2M2_NARROW_MT")) : 0;       // A/B switch: 1 = 4-row blocks, 2 = 8-row blocks
    const long long blocks8 = (long long)a.N * ((a.Wo + 31) / 32) * ((a.Ho + 7) / 8);
    return force_mt ? force_mt == 2 : blocks8 >= 1024;
}

}  // namespace s2m2

extern "C" int s2m2_conv_narrow_supported(int KH, int KW, int stride, int Cin, int Cout, int dtype) {
    if (dtype != S2M2_F16 || Cout <= 0 || Cout % 8) return 0;
    if (KH == 3 && KW == 3 && stride == 1 && Cin == 8) return 1;
    if (KH == 5 && KW == 5 && stride == 2 && Cin == 16) return ((Cout + 31) / 32) % 2 == 0;          // two cout tiles per group
    if (KH == 3 && KW == 3 && stride == 1) {                       // pixel-split form on wider inputs: few output channels
        if (Cin == 48) return Cout <= 64;
        if (Cin == 96) return Cout <= 96;
        if (Cin == 128 || Cin == 256) return Cout <= 32;
    }
    return 0;
}

static int conv_narrow_impl(const s2m2_narrow_desc* d, void* stream) {
    using namespace s2m2;
    S2M2_REQUIRE(d, "conv_narrow: null descriptor");
    S2M2_REQUIRE(d->x && d->weight_frag && d->out, "conv_narrow: null pointer");
    S2M2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && (long long)d->N * d->H * d->W < (1LL << 31), "conv_narrow: N=%d H=%d W=%d", d->N, d->H, d->W);
    S2M2_REQUIRE(s2m2_conv_narrow_supported(d->KH, d->KW, d->stride, d->Cin, d->Cout, d->dtype),
                 "conv_narrow: %dx%d stride %d Cin=%d Cout=%d dtype=%d is not supported (ask s2m2_conv_narrow_supported)", d->KH, d->KW, d->stride,
                 d->Cin, d->Cout, d->dtype);
    S2M2_REQUIRE(d->head_cout >= 0 && (d->head_cout == 0 || (d->Cin == 48 && d->head_frag && d->head_cout % 8 == 0 && d->head_cout <= 32)),
                 "conv_narrow: the fused 1x1 head (head_cout=%d) exists for the 48-channel form, with a fragment tensor and at most 32 output channels", d->head_cout);
    S2M2_REQUIRE(d->head_cout == 0 || d->act == S2M2_ACT_NONE || d->act == S2M2_ACT_RELU, "conv_narrow: the fused 1x1 head takes act NONE or RELU (act=%d)", d->act);
    S2M2_REQUIRE(d->x_stride >= d->Cin - d->Cin1 && d->x_stride % 8 == 0 && d->out_stride >= (d->head_cout ? d->head_cout : d->Cout) && d->out_stride % 8 == 0,
                 "conv_narrow: x_stride=%lld / out_stride=%lld must cover the channels and be multiples of 8", d->x_stride, d->out_stride);
    S2M2_REQUIRE(d->act == S2M2_ACT_NONE || d->act == S2M2_ACT_GELU || d->act == S2M2_ACT_RELU, "conv_narrow: act=%d (NONE, GELU or RELU)", d->act);
    S2M2_REQUIRE(d->Cin1 >= 0 && d->Cin1 < d->Cin && d->Cin1 % 8 == 0 && (d->Cin1 == 0 || (d->x1 && d->x1_stride >= d->Cin1 && d->x1_stride % 8 == 0)),
                 "conv_narrow: the second source needs a pointer, Cin1=%d a multiple of 8 below Cin and a pixel stride that is a multiple of 8", d->Cin1);
    S2M2_REQUIRE(d->Cin1 == 0 || d->Cin > 16, "conv_narrow: the 8- / 16-channel forms read one source");
    NarrowArgs a;
    a.x = d->x; a.xstride = d->x_stride; a.N = d->N; a.H = d->H; a.W = d->W;
    a.c0 = d->Cin - d->Cin1; a.Cin = d->Cin;
    a.x1 = d->Cin1 ? d->x1 : d->x; a.x1stride = d->Cin1 ? d->x1_stride : d->x_stride;
    a.nchunk = d->Cin >= 128 ? d->Cin / 64 : 1;
    a.Ho = (d->H + d->stride - 1) / d->stride; a.Wo = (d->W + d->stride - 1) / d->stride;
    a.w = d->weight_frag; a.bias = d->bias; a.out = d->out; a.out_stride = d->out_stride; a.Cout = d->Cout; a.act = d->act;
    a.w2 = d->head_frag; a.bias2 = d->head_bias; a.Cout2 = d->head_cout;
    a.zero = zero_page();
    S2M2_REQUIRE(a.zero, "conv_narrow: cannot allocate the zero page");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool tall = narrow_tall(a);
    if (d->Cin == 48 && a.Cout2) return tall ? launch_px<48, 2, 2, 4, true>(a, st) : launch_px<48, 2, 1, 4, true>(a, st);
    if (d->Cin == 48) return tall ? launch_px<48, 2, 2>(a, st) : launch_px<48, 2, 1>(a, st);
    // measured (profiles/r04/narrowbench.txt, narrowbench_variants.txt): 96 -> 96 with ONE cout tile per block (grid.y = 3) 27.2 us against 27.9 for
    // all three in one block; the 128- / 256-channel heads in chunks of 64 channels with 2 waves x 2 rows per block 22.7 / 12.5 us against
    // 25.1 / 13.4 (4 waves x 1 row) and 27.4 / 14.3 (chunks of 128); 1-wave blocks 27 - 45 us.  Counters (profiles/r04/narrow_pmc.txt): the
    // TCP serves ~80 % of the fragment requests the waves of a block repeat; these launches are short chains of tile loads, barriers and K
    // loops of 36 - 54 MFMAs on a chip their 640 blocks fill a quarter to a half -- not MFMA-, HBM- or L2-bound.
    if (d->Cin == 96) return launch_px<96, 1, 1>(a, st);
    if (d->Cin >= 128) return launch_px<64, 1, 2, 2>(a, st);
    if (d->Cin == 8) return tall ? launch_narrow<3, 3, 1, 8, 2, 1, 1>(a, st) : launch_narrow<3, 3, 1, 8, 1, 1, 1>(a, st);
    // 5x5: wave pairs split the two cout tiles of a group and share 2 / 4 output rows (A/B switch S2M2_NARROW_NWN=1: every wave both tiles)
    static const bool split = !(getenv("S2M2_NARROW_NWN") && atoi(getenv("S2M2_NARROW_NWN")) == 1);
    if (split) return tall ? launch_narrow<5, 5, 2, 16, 4, 1, 2>(a, st) : launch_narrow<5, 5, 2, 16, 2, 1, 2>(a, st);
    return tall ? launch_narrow<5, 5, 2, 16, 2, 2, 1>(a, st) : launch_narrow<5, 5, 2, 16, 1, 2, 1>(a, st);
}
extern "C" int s2m2_conv_narrow(const s2m2_narrow_desc* d, void* stream) {
    return s2m2::plan_dispatch_desc<s2m2_narrow_desc>("s2m2_conv_narrow", &conv_narrow_impl, d, stream);
}

