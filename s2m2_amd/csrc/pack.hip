// Weight packing into the MFMA-fragment orders of the direct-form kernels (K5 v5, K9, K10, K11, K12): s2m2_pack_frag.
//
// The direct forms read their weights as streams of 1 KB MFMA A-fragments in exactly the order a wave consumes them (one 16-byte piece per
// lane: lane l of the fragment of 32-row tile t and k16 step s holds row 32 t + l % 32, columns 16 s + 8 (l / 32) + 0..7).  Up to ABI 400 the
// permutations lived in the Python binding (s2m2_amd/pack.py); a caller of the C ABI had to re-derive them.  Here they are part of the library:
// the caller hands over the plain packing -- (Cout_padded, K) row-major fp16, K = (tap, channel) with channel fastest, what
// nn.Conv2d.weight.permute(0, 2, 3, 1).reshape(Cout, -1) gives -- and gets the stream the kernel wants.  One gather kernel serves every
// order: destination element (tile t, step S, lane l, e) <- src[32 t + l % 32][colmap[16 S + 8 (l / 32) + e]] (0 where the map says -1 or the
// row is past the source); the column map of an order is built on the host.  Packing is a one-time, synchronous set-up step (it allocates and
// frees its map): not for use under stream capture.
#include "common.h"
#include <vector>

namespace s2m2 {

__global__ __launch_bounds__(256) void frag_pack_kernel(const half_t* __restrict__ src, long long src_ld, int src_rows, const int* __restrict__ colmap,
                                                        half_t* __restrict__ dst, long long tile_stride, long long step0, int ntiles, int nsteps) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;          // one 16-byte piece per thread
    const long long total = (long long)ntiles * nsteps * 64;
    if (gid >= total) return;
    const int lane = (int)(gid & 63);
    const long long ts = gid >> 6;
    const int s = (int)(ts % nsteps);
    const int t = (int)(ts / nsteps);
    const int row = 32 * t + (lane & 31);
    const int* cm = colmap + (size_t)s * 16 + 8 * (lane >> 5);
    alignas(16) half_t v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cm[e];
        v[e] = (row < src_rows && c >= 0) ? src[(size_t)row * src_ld + c] : (half_t)0.f;
    }
    *reinterpret_cast<raw16_t*>(dst + ((size_t)t * tile_stride + (size_t)(step0 + s) * 64 + lane) * 8) = *reinterpret_cast<const raw16_t*>(v);
}

struct PackPlan {                         // one gather launch
    const half_t* src; long long src_ld; int src_rows;
    std::vector<int> colmap;              // 16 entries per step
    long long tile_stride, step0; int ntiles, nsteps;
};

static int run_plan(const PackPlan& p, half_t* dst, hipStream_t st) {
    int* dmap = nullptr;
    const size_t bytes = p.colmap.size() * sizeof(int);
    if (hipMalloc(&dmap, bytes) != hipSuccess) return set_error("pack_frag: cannot allocate the column map (%zu bytes)", bytes);
    int rc = 0;
    if (hipMemcpy(dmap, p.colmap.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = set_error("pack_frag: column map upload failed");
    if (!rc) {
        const long long total = (long long)p.ntiles * p.nsteps * 64;
        hipLaunchKernelGGL(frag_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p.src, p.src_ld, p.src_rows, dmap, dst,
                           p.tile_stride, p.step0, p.ntiles, p.nsteps);
        rc = check_launch("pack_frag");
        if (!rc && hipStreamSynchronize(st) != hipSuccess) rc = set_error("pack_frag: kernel failed");
    }
    (void)hipFree(dmap);
    return rc;
}

static std::vector<int> identity_map(int nsteps, int k, int offset = 0) {
    std::vector<int> m((size_t)nsteps * 16);
    for (int c = 0; c < nsteps * 16; ++c) m[c] = c < k ? offset + c : -1;
    return m;
}

}  // namespace s2m2

// elements (fp16) of the packed stream for this descriptor; < 0: bad descriptor (s2m2_last_error)
extern "C" long long s2m2_pack_frag_elems(const s2m2_pack_desc* d) {
    using namespace s2m2;
    if (!d) { set_error("pack_frag: null descriptor"); return -1; }
    if (d->rows <= 0 || d->cols <= 0) { set_error("pack_frag: rows=%d cols=%d", d->rows, d->cols); return -1; }
    const long long tiles = (d->rows + 31) / 32;
    switch (d->kind) {
        case S2M2_PACK_ROWS:
        case S2M2_PACK_NARROW: return tiles * 32 * ((d->cols + 15) / 16 * 16);
        case S2M2_PACK_CONV_FRAG: {
            if (d->ntap <= 0 || d->cols % d->ntap) { set_error("pack_frag: cols=%d is not ntap=%d taps of channels", d->cols, d->ntap); return -1; }
            const int cin = d->cols / d->ntap;
            const int ck = s2m2_conv_frag_chunk(d->rows, cin);
            return tiles * 32 * (long long)d->ntap * ((cin + ck - 1) / ck * ck);
        }
        case S2M2_PACK_FUSION: return 9LL * d->rows * d->rows;
        case S2M2_PACK_HEAD: return 64LL * 8 * 2 * ((d->cols + 31) / 32);
        default: set_error("pack_frag: unknown kind %d", d->kind); return -1;
    }
}

extern "C" int s2m2_pack_frag(const s2m2_pack_desc* d, void* stream) {
    using namespace s2m2;
    const long long need = s2m2_pack_frag_elems(d);
    if (need < 0) return 1;
    S2M2_REQUIRE(d->w && d->out, "pack_frag: null pointer");
    S2M2_REQUIRE(d->out_elems >= need, "pack_frag: out holds %lld elements, the stream needs %lld", d->out_elems, need);
    S2M2_REQUIRE(d->ld == 0 || d->ld >= d->cols, "pack_frag: ld=%d below cols=%d", d->ld, d->cols);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const half_t* w = static_cast<const half_t*>(d->w);
    half_t* out = static_cast<half_t*>(d->out);
    const long long ld = d->ld ? d->ld : d->cols;
    const int tiles = (d->rows + 31) / 32;
    PackPlan p;
    p.src = w; p.src_ld = ld; p.src_rows = d->rows; p.ntiles = tiles; p.step0 = 0;
    switch (d->kind) {
        case S2M2_PACK_ROWS: {                       // K9 / K11: [tile][step][lane][8], zero padded to whole tiles and steps
            p.nsteps = (d->cols + 15) / 16;
            p.colmap = identity_map(p.nsteps, d->cols);
            p.tile_stride = (long long)p.nsteps * 64;
            return run_plan(p, out, st);
        }
        case S2M2_PACK_NARROW: {                     // K12: as ROWS; layers with >= 128 input channels run as chunks of 64 channels, K = (chunk, tap, channel in chunk)
            S2M2_REQUIRE(d->ntap > 0 && d->cols % d->ntap == 0, "pack_frag: cols=%d is not ntap=%d taps of channels", d->cols, d->ntap);
            const int cin = d->cols / d->ntap;
            p.nsteps = (d->cols + 15) / 16;
            p.colmap = identity_map(p.nsteps, d->cols);
            if (cin >= 128) {
                S2M2_REQUIRE(cin % 64 == 0, "pack_frag: narrow layers with >= 128 input channels need a multiple of 64 (Cin=%d)", cin);
                for (int k = 0; k < d->cols; ++k) {
                    const int chunk = k / (d->ntap * 64), r = k - chunk * d->ntap * 64, tap = r / 64, c = r - tap * 64;
                    p.colmap[k] = tap * cin + chunk * 64 + c;
                }
            }
            p.tile_stride = (long long)p.nsteps * 64;
            return run_plan(p, out, st);
        }
        case S2M2_PACK_CONV_FRAG: {                  // K5 v5 (K order 2): [tile][chunk of 128 / 192 channels][tap][k16 step], zero beyond Cin
            S2M2_REQUIRE(d->rows % 32 == 0, "pack_frag: K order 2 needs Cout=%d to be a multiple of 32", d->rows);
            const int cin = d->cols / d->ntap, ck = s2m2_conv_frag_chunk(d->rows, cin), nchunk = (cin + ck - 1) / ck, nks = ck / 16;
            p.nsteps = nchunk * d->ntap * nks;
            p.colmap.assign((size_t)p.nsteps * 16, -1);
            for (int ch = 0; ch < nchunk; ++ch)
                for (int tap = 0; tap < d->ntap; ++tap)
                    for (int ks = 0; ks < nks; ++ks)
                        for (int q = 0; q < 16; ++q) {
                            const int c = ch * ck + ks * 16 + q;
                            if (c < cin) p.colmap[((size_t)(ch * d->ntap + tap) * nks + ks) * 16 + q] = tap * cin + c;
                        }
            p.tile_stride = (long long)p.nsteps * 64;
            return run_plan(p, out, st);
        }
        case S2M2_PACK_FUSION: {                     // K10: per 32-cout tile, for slice s = 0, 1, 2: W1[sC + 32t .., :] (2C/16 steps), then W2[32t .., sC:(s+1)C] (C/16 steps)
            const int C = d->rows;
            S2M2_REQUIRE(d->w2 && C % 32 == 0 && d->cols == 2 * C, "pack_frag: fusion needs w (3C, 2C) given as rows = C, cols = 2C, and w2 (C, 3C)");
            const int ks0 = 2 * C / 16, ks1 = C / 16, per = ks0 + ks1;
            const half_t* w2 = static_cast<const half_t*>(d->w2);
            const long long ld2 = d->ld2 ? d->ld2 : 3LL * C;
            for (int s = 0; s < 3; ++s) {
                PackPlan a;
                a.src = w + (size_t)s * C * ld; a.src_ld = ld; a.src_rows = C; a.ntiles = C / 32; a.nsteps = ks0;
                a.colmap = identity_map(ks0, 2 * C); a.tile_stride = 3LL * per * 64; a.step0 = (long long)s * per;
                if (run_plan(a, out, st)) return 1;
                PackPlan b;
                b.src = w2; b.src_ld = ld2; b.src_rows = C; b.ntiles = C / 32; b.nsteps = ks1;
                b.colmap = identity_map(ks1, C, s * C); b.tile_stride = 3LL * per * 64; b.step0 = (long long)s * per + ks0;
                if (run_plan(b, out, st)) return 1;
            }
            return 0;
        }
        default: {                                   // S2M2_PACK_HEAD: the 1x1 layer fused behind a K12 3x3 layer (s2m2_narrow_desc.head_frag)
            S2M2_REQUIRE(d->rows <= 32, "pack_frag: a fused head has at most 32 output channels (rows=%d)", d->rows);
            const int nj = (d->cols + 31) / 32;
            p.ntiles = 1; p.nsteps = 2 * nj; p.tile_stride = (long long)p.nsteps * 64;
            p.colmap.assign((size_t)p.nsteps * 16, -1);
            // step (j, p), half h, element 4 q + e <- channel 32 j + 8 (2 p + q) + 4 h + e: what lane (pixel, h) of the 3x3 layer's accumulator
            // tile j holds in its register quads 2 p and 2 p + 1
            for (int j = 0; j < nj; ++j)
                for (int pp = 0; pp < 2; ++pp)
                    for (int h = 0; h < 2; ++h)
                        for (int q = 0; q < 2; ++q)
                            for (int e = 0; e < 4; ++e) {
                                const int c = 32 * j + 8 * (2 * pp + q) + 4 * h + e;
                                if (c < d->cols) p.colmap[((size_t)(j * 2 + pp)) * 16 + 8 * h + 4 * q + e] = c;
                            }
            return run_plan(p, out, st);
        }
    }
}
