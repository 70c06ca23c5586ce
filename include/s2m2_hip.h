/*
 * s2m2_hip.h -- C ABI of libs2m2_hip.so: hand-written CDNA4 (gfx950) kernels for the S2M2 inference hot path.
 *
 * The reference (junhong-3dv/s2m2) has no native/FFI layer: its hot path is the body of
 * S2M2.forward (src/s2m2/core/model/s2m2.py:136-197) expressed as ATen calls.  Every entry point below
 * replaces one group of those call sites (cited per function, SURVEY.md section 8a ids in brackets) and is what a
 * maintainer would bind from Python with ctypes.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch); nothing is allocated inside;
 *   - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); all work is enqueued on it,
 *     no synchronisation, safe to capture in a hipGraph;
 *   - dtype codes: S2M2_F32 = 0, S2M2_F16 = 1;
 *   - return 0 on success, non-zero on error; s2m2_last_error() returns a thread-local message;
 *   - image-like activations are channels-last: (N, H, W, C) with C fastest ("NHWC"), tokens are rows;
 *   - thread-safe for distinct streams.
 */
#ifndef S2M2_HIP_H
#define S2M2_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { S2M2_F32 = 0, S2M2_F16 = 1 };

/* library version (major*10000 + minor*100 + patch) and last error text of the calling thread */
int s2m2_version(void);
const char* s2m2_last_error(void);

/* Name of the device kernel s2m2_ln_corr dispatches to for this configuration (for rocprof matching). */
const char* s2m2_ln_corr_kernel_name(int C, int feat_dtype, int cv_dtype);

/*
 * [A4] LayerNorm + all-pairs epipolar correlation  (DispInit.forward, submodules.py:216-217; LayerNorm :165)
 *   feat   (2B, h, w, C) NHWC, left images = batch entries [0,B), right = [B,2B)   dtype feat_dtype
 *   ln_w, ln_b  (C) fp32   LayerNorm affine (eps 1e-5, biased variance)
 *   cv     (B, h, w, w)  cv[b,y,i,j] = < LN(feat[b,y,i,:]), LN(feat[B+b,y,j,:]) >   dtype cv_dtype, j fastest
 *   C in {64,128,192,256,384}; w % 8 == 0.  F16: LN in fp32, operands rounded to fp16, fp32 accumulate (MFMA);
 *   F32: exact fp32 MFMA (v_mfma_f32_32x32x2_f32).
 */
int s2m2_ln_corr(const void* feat, const float* ln_w, const float* ln_b, void* cv,
                 int B, int h, int w, int C, int feat_dtype, int cv_dtype, void* stream);

/*
 * [A5+A6] Sinkhorn optimal transport with dustbins + argmax + 5-tap window regression
 *   (DispInit._optimal_transport/_sinkhorn submodules.py:169-201, regression :225-241, logsumexp_stable :147-152)
 *   cv      (B, h, w, w) dtype cv_dtype (read only)
 *   disp, conf, occ  (B, h, w) fp32 out  (disp = i - soft-argmax, 1/4-res pixels; occ = row mass of masked P)
 *   argmax  (B, h, w) int32 out or NULL  (first maximal j wins)
 *   use_positivity: entries j > i are masked (the reference fills -1e4, which underflows to exactly 0 in fp32)
 *   workspace: NULL or >= s2m2_sinkhorn_workspace_bytes(...) bytes.
 */
size_t s2m2_sinkhorn_workspace_bytes(int B, int h, int w, int cv_dtype);
int s2m2_sinkhorn_regress(const void* cv, float* disp, float* conf, float* occ, int32_t* argmax,
                          int B, int h, int w, int ot_iter, int use_positivity, int cv_dtype,
                          void* workspace, void* stream);

/*
 * [A9+A10] two-level cost-volume lookup, radius r (CostVolume.__init__/__call__, submodules.py:23-60,
 *   bilinear_sampler :7-17).  Level 1 (cv averaged over pairs of j) is computed on the fly, never stored.
 *   cv    (B, h, w, w) dtype cv_dtype;  disp (B, h, w) fp32
 *   corr1, corr2  fp32 (or fp16 when out_dtype = S2M2_F16) with element (b,y,i,k) at
 *       base + ((b*h + y)*w + i)*pix_stride + k*tap_stride      k = 0..2r  <->  dx = k - r
 *   (planar (B,2r+1,h,w) of the reference: pix_stride=1, tap_stride=h*w with base offset b*(2r+1)*h*w handled by
 *    batch_stride; channels-last: pix_stride=2r+1 (or larger), tap_stride=1)
 *   The reference's pixel -> normalised -> pixel fp32 coordinate round trip (grid_sample, align_corners=True,
 *   zeros padding) is reproduced on both axes.
 */
int s2m2_cv_lookup(const void* cv, const float* disp, void* corr1, void* corr2,
                   int B, int h, int w, int radius, int cv_dtype, int out_dtype,
                   long long batch_stride, long long pix_stride, long long tap_stride, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* S2M2_HIP_H */
