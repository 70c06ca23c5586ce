// One tap of the two-level cost-volume lookup (K3; reference submodules.py:7-60), shared by cv_lookup.hip and by
// the fused lookup + corr_feat kernel of pw.hip.  The reference goes pixel -> normalised (Python, `2*x/(W-1)-1`) -> pixel (ATen CPU kernel,
// `(g+1)*((size-1)/2)`) in fp32 on BOTH axes, so the effective coordinate is off by ~1 ulp and a vanishing weight can land on the
// neighbouring row; that arithmetic is reproduced here with contraction disabled.
#pragma once
#include "common.h"

namespace s2m2 {

#pragma clang fp contract(off)
__device__ __forceinline__ float roundtrip(float pix, float size) {
    const float g = 2.0f * pix / (size - 1.0f) - 1.0f;     // reference Python (submodules.py:12-13)
    return (g + 1.0f) * ((size - 1.0f) / 2.0f);            // ATen CPU unnormalize, align_corners=True
}

template <typename TI, int LEVEL>
__device__ __forceinline__ float fetch(const TI* __restrict__ row, int x, int ws) {
    // value of the sampled image at column x of this row; zeros padding outside [0, ws-1]
    if (x < 0 || x >= ws) return 0.f;
    if (LEVEL == 0) return to_f32(row[x]);
    return (to_f32(row[2 * x]) + to_f32(row[2 * x + 1])) * 0.5f;
}

// img: the (w x w, rows `pitch` apart) slice cv[b, y] of left pixel row (b, y); i: left pixel column; d: its disparity; level 0 / 1; tap k
template <typename TI>
__device__ __forceinline__ float lookup_tap(const TI* __restrict__ img, int i, float d, int level, int k, int radius, int w, int pitch) {
    const float dx = (float)(k - radius);
    float x, wsf;
    int ws;
    if (level == 0) { x = ((float)i - d) + dx; ws = w; }
    else            { x = ((float)i / 2.0f - d / 2.0f) + dx; ws = w / 2; }
    wsf = (float)ws;
    const float ix = roundtrip(x, wsf);
    const float iy = roundtrip((float)i, (float)w);
    const float x0f = floorf(ix), y0f = floorf(iy);
    const float wx = ix - x0f, wy = iy - y0f;
    const float ex = 1.0f - wx, ey = 1.0f - wy;
    // far out-of-range coordinates (huge |d|): every tap is zero padding
    float out = 0.f;
    if (x0f >= -2.0f && x0f <= wsf + 1.0f) {
        const int x0 = (int)x0f, y0 = (int)y0f;
        const float w00 = ey * ex, w01 = ey * wx, w10 = wy * ex, w11 = wy * wx;
        if (y0 >= 0 && y0 < w) {
            const TI* r0 = img + (size_t)y0 * pitch;
            const float a = level ? fetch<TI, 1>(r0, x0, ws) : fetch<TI, 0>(r0, x0, ws);
            const float c = level ? fetch<TI, 1>(r0, x0 + 1, ws) : fetch<TI, 0>(r0, x0 + 1, ws);
            out = out + a * w00;
            out = out + c * w01;
        }
        if (y0 + 1 >= 0 && y0 + 1 < w && wy != 0.0f) {
            const TI* r1 = img + (size_t)(y0 + 1) * pitch;
            const float a = level ? fetch<TI, 1>(r1, x0, ws) : fetch<TI, 0>(r1, x0, ws);
            const float c = level ? fetch<TI, 1>(r1, x0 + 1, ws) : fetch<TI, 0>(r1, x0 + 1, ws);
            out = out + a * w10;
            out = out + c * w11;
        }
    }
    return out;
}
#pragma clang fp contract(on)

}  // namespace s2m2
