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

/* ABI version of THIS header (major*10000 + minor*100 + patch).  s2m2_version() returns the value the library was built with: a caller
 * compares the two before its first call (s2m2_amd/hip.py: load() refuses a library whose value differs from the binding's: descriptor
 * layouts change IN PLACE between patch versions too, and a stale git-ignored .so would read pointers at shifted offsets).  The value is
 * bumped with EVERY change of a signature or descriptor layout.
 * History: 100 = rounds 1-2; 300 = round 3 changed signatures IN PLACE (cv_pitch inserted into s2m2_sinkhorn_regress / s2m2_cv_lookup,
 * s2m2_conv_desc / s2m2_chain_desc grew epi_cout0, ln_out*, fan_*, weight_frag, pool_h / pool_w) -- a caller built against 100 must be
 * rebuilt; 400 = round 4 (s2m2_pw_direct, s2m2_conv_narrow, s2m2_ln_corr_pitched added; the round-3 experiment entry
 * points s2m2_corr_tiled / s2m2_corr_hybrid / s2m2_debug_store_pattern and the ln_out_tile* fields of s2m2_chain_desc removed;
 * head_* appended to s2m2_narrow_desc without a bump -- the reason for the exact comparison since 500); 500 = round 5; 600 = round 6
 * (s2m2_row_attn and s2m2_conv_block added; the five ABI-400 entry points of K1 -- s2m2_ln_corr, _timed, _banded, _pitched, s2m2_corr -- removed: every form of
 * K1 is s2m2_cost_volume). */
#define S2M2_ABI_VERSION 600
int s2m2_version(void);
const char* s2m2_last_error(void);
/* test aid (not part of the path): fills the LDS of every CU with quiet-NaN patterns, so that a kernel launched next that reads an LDS word it
 * never wrote produces NaNs instead of silently using stale data (tests/test_hip_lds_poison.py) */
int s2m2_debug_poison_lds(void* stream);
/* measurement aid (not part of the path): writes {shader-clock ticks (s_memtime), 100 MHz real-time ticks (s_memrealtime)} as two
 * uint64 to device memory `out`; two probes around a region of a stream give the average engine clock it ran at (tools/clock_probe.py) */
int s2m2_debug_clock_probe(void* out, void* stream);

/* Name of the device kernel s2m2_cost_volume dispatches to for this configuration (for rocprof matching). */
const char* s2m2_ln_corr_kernel_name(int C, int feat_dtype, int cv_dtype);

/*
 * [A4] K1: (LayerNorm +) all-pairs epipolar correlation = the cost volume  (DispInit.forward, submodules.py:216-217; LayerNorm :165)
 *   ONE descriptor for every form of the kernel (ABI 500; the five positional entry points of ABI 400 were removed with ABI 600):
 *   tokens  (2B, h, w, C) NHWC, left images = batch entries [0,B), right = [B,2B)            dtype token_dtype
 *   ln_weight, ln_bias  (C) fp32: LayerNorm affine (eps 1e-5, biased variance), applied INSIDE the kernel.  Both NULL: the tokens are
 *           normalised already (DispInit's layer_norm folded into the launch that produced them: the ln_out rows of s2m2_mlp_chain) and
 *           the kernel is the batched product R . L^T and its stores alone -- the shipped path at C = 128 / 192 / 256 / 384 in fp16
 *   cv      cv[b,y,i,j] = < LN(tokens[b,y,i,:]), LN(tokens[B+b,y,j,:]) >  at  cv + ((b*h + y)*w + i)*cv_pitch + j       dtype cv_dtype
 *   cv_pitch  elements between volume rows, >= w, a multiple of 8 (0 = w).  A multiple of 64 fp16 / 32 fp32 elements (w = 304 -> 320) makes
 *           every 64-column store granule a whole 128-byte line (then the stores are write-through, sc1); s2m2_sinkhorn_regress and
 *           s2m2_cv_lookup read the same pitch
 *   band    -1: the full volume.  >= 0 (use_positivity models): only columns j <= i + band are written (rounded up to the
 *           64-column store granule), the rest of the buffer is left untouched -- nothing downstream reads beyond band = 11: the Sinkhorn
 *           mask removes j > i (submodules.py:211-214), the +-4 tap lookups at both pyramid levels reach column i + 11 (submodules.py:39-60)
 *   start_event, stop_event  optional hipEvent_t (s2m2_event_create): recorded on the kernel dispatch itself (hipExtLaunchKernel), i.e. they
 *           bracket the kernel's execution like a rocprofv3 kernel trace does, not the gaps around it (bench.py `roofline`); elapsed time
 *           is valid once the stream has been synchronised
 *   C in {64,128,192,256,384}; w % 8 == 0.  F16: LN in fp32, operands rounded to fp16, fp32 accumulate (MFMA); F32: exact fp32 MFMA
 *   (v_mfma_f32_32x32x2_f32).  Pairs (token_dtype, cv_dtype): (F16,F16), (F16,F32), (F32,F32).
 */
typedef struct s2m2_corr_desc {
    const void* tokens;
    const float* ln_weight;
    const float* ln_bias;
    void* cv;
    int B, h, w, C;
    int cv_pitch;
    int band;
    int token_dtype, cv_dtype;
    void* start_event;
    void* stop_event;
} s2m2_corr_desc;
int s2m2_cost_volume(const s2m2_corr_desc* desc, void* stream);

/*
 * Weight packing into the MFMA-fragment orders of the direct-form kernels (ABI 500; up to ABI 400 these permutations lived in the Python
 * binding, s2m2_amd/pack.py, and a C caller had to re-derive them).  Input: the PLAIN packing of a layer -- (rows = Cout padded to 8, cols = K)
 * row-major fp16 with K = (tap, channel), channel fastest, every source's channel count padded to 8 (what
 * conv.weight.permute(0, 2, 3, 1).reshape(Cout, -1) gives for an unpadded layer; ld = elements between rows, 0 = cols).  Output: the stream the
 * kernel consumes, s2m2_pack_frag_elems() fp16 elements (-1: bad descriptor):
 *   S2M2_PACK_ROWS       s2m2_chain_desc.weight / fan_weight with weight_frag = 1 (K9; stack the layers of fan_weight along rows),
 *                        s2m2_pw_desc.weight_frag (K11): [row tile of 32][k16 step][lane][8], zero padded to whole tiles / steps
 *   S2M2_PACK_NARROW     s2m2_narrow_desc.weight_frag (K12), ntap = KH * KW: as ROWS; layers on >= 128 input channels in chunks of 64
 *   S2M2_PACK_CONV_FRAG  s2m2_conv_desc.weight with korder = 2 (K5 v5), ntap = KH * KW: [cout tile][channel chunk][tap][k16 step], chunks of
 *                        s2m2_conv_frag_chunk(rows, cols / ntap) channels (128; 192 for the layers of 192-channel models)
 *   S2M2_PACK_FUSION     s2m2_feature_fusion_frag's stream (K10): w = the first layers [feature_gate.0 ; feature_fusion.0] (3C, 2C) given as
 *                        rows = C, cols = 2C, w2 = [feature_gate.2 | feature_fusion.2] (C, 3C) (ld2: its row stride, 0 = 3C)
 *   S2M2_PACK_HEAD       s2m2_narrow_desc.head_frag: the 1x1 layer (rows <= 32 couts, cols = the 3x3 layer's couts) fused behind a K12 layer
 * One-time synchronous set-up (allocates and frees a small index map, waits for the stream): not for use under stream capture.
 */
enum { S2M2_PACK_ROWS = 0, S2M2_PACK_NARROW = 1, S2M2_PACK_CONV_FRAG = 2, S2M2_PACK_FUSION = 3, S2M2_PACK_HEAD = 4 };
/* channels per chunk of the K-order-2 stream of a layer (one input patch in LDS per chunk): 128, or 192 where that divides both sides and 128 does
 * not divide Cout (Cout = Cin = 192, 192 <- 384: blocks of 192 couts, no padded couts, no half-empty chunk).  Packers and s2m2_conv2d use this rule. */
static inline int s2m2_conv_frag_chunk(int cout, int cin) { return (cout % 128 != 0 && cout % 192 == 0 && cin % 192 == 0) ? 192 : 128; }
typedef struct s2m2_pack_desc {
    int kind;
    const void* w;
    const void* w2;
    int rows, cols;
    int ld, ld2;
    int ntap;
    void* out;
    long long out_elems;
} s2m2_pack_desc;
long long s2m2_pack_frag_elems(const s2m2_pack_desc* desc);
int s2m2_pack_frag(const s2m2_pack_desc* desc, void* stream);

/*
 * Recorded launch plans (ABI 500): a sequence of library calls replayed from C.  Between s2m2_plan_begin and s2m2_plan_end every launch-type entry
 * point the CALLING THREAD invokes is executed as always AND appended to the plan (arguments by value, descriptors copied).  s2m2_plan_end
 * declares the plan's EXTERNAL buffers -- next ranges [ext_base[i], ext_base[i] + ext_bytes[i]) of device memory: every recorded pointer into
 * range i is stored relative to it (ABI 600: only the pointer-typed arguments and the pointer fields of the descriptors are compared with the
 * ranges -- a size, a stride or a pair of ints that falls inside a range is never rewritten) -- and s2m2_plan_run re-issues the whole sequence
 * on `stream` with the externals at ext_ptrs[i] (same count
 * and order; a NULL external is accepted if no recorded call points into it).  All other pointers (weights, scratch, the intermediate tensors
 * of the recorded run) are replayed as recorded: the owner of the plan keeps those allocations alive and unshared for the life of the plan.
 * A plan is immutable once sealed; concurrent runs on different streams are safe when their externals differ and the caller accepts that the
 * internal intermediates are shared (i.e. in practice: one run at a time per plan).  Recording does not synchronise and may itself run under
 * stream capture; s2m2_plan_run may too.  s2m2_plan_abort ends a recording whose owner failed half way (the plan can only be destroyed then).
 *
 * s2m2_refine_step: ONE refinement iteration -- LocalRefiner.forward + the loop epilogue of S2M2.forward (refinenet.py:126-154, s2m2.py:175-180):
 * cost-volume lookup, the correlation / disparity / confidence feature layers, the refinement U-Net with its attention blocks, the ConvGRU, the
 * update heads, refine_update; about 55 launches -- as one native call: a plan recorded around that iteration whose seven externals are, in this
 * order, hidden (B,h,w,C), ctx (B,h,w,C), disp, conf, occ (B,1,h,w fp32), cv (B,h,w,cv_pitch) and the (B,h,w,8) side input written by the previous
 * iteration's epilogue (NULL for the first iteration, which builds its own).  The iteration's outputs are the tensors of the recorded run
 * (s2m2_amd/engine.py keeps them; a C caller records its own plan around its own launch sequence the same way).
 */
typedef struct s2m2_plan s2m2_plan;
int s2m2_plan_begin(s2m2_plan** plan);
int s2m2_plan_end(s2m2_plan* plan, const void* const* ext_base, const size_t* ext_bytes, int next);
int s2m2_plan_abort(s2m2_plan* plan);
int s2m2_plan_launches(const s2m2_plan* plan);
int s2m2_plan_patches(const s2m2_plan* plan, int slot);   /* recorded pointers that follow external `slot` (< 0: all): diagnostics */
int s2m2_plan_run(const s2m2_plan* plan, const void* const* ext_ptrs, int next, void* stream);
int s2m2_plan_destroy(s2m2_plan* plan);
int s2m2_refine_step(const s2m2_plan* step, const void* hidden, const void* ctx, const void* disp, const void* conf, const void* occ,
                     const void* cv, const void* side_input, void* stream);

int s2m2_event_create(void** event);
int s2m2_event_destroy(void* event);
int s2m2_event_elapsed_us(void* start_event, void* stop_event, float* microseconds);

/*
 * [A5+A6] Sinkhorn optimal transport with dustbins + argmax + 5-tap window regression
 *   (DispInit._optimal_transport/_sinkhorn submodules.py:169-201, regression :225-241, logsumexp_stable :147-152)
 *   cv      (B, h, w, w) dtype cv_dtype (read only), volume rows cv_pitch elements apart (0 = w; a multiple of 8)
 *   disp, conf, occ  (B, h, w) fp32 out  (disp = i - soft-argmax, 1/4-res pixels; occ = row mass of masked P)
 *   argmax  (B, h, w) int32 out or NULL  (first maximal j wins)
 *   use_positivity: entries j > i are masked (the reference fills -1e4, which underflows to exactly 0 in fp32)
 *   workspace: NULL or >= s2m2_sinkhorn_workspace_bytes(...) bytes.
 */
size_t s2m2_sinkhorn_workspace_bytes(int B, int h, int w, int cv_dtype);
int s2m2_sinkhorn_regress(const void* cv, float* disp, float* conf, float* occ, int32_t* argmax,
                          int B, int h, int w, int ot_iter, int use_positivity, int cv_dtype, int cv_pitch,
                          void* workspace, void* stream);

/*
 * [A9+A10] two-level cost-volume lookup, radius r (CostVolume.__init__/__call__, submodules.py:23-60,
 *   bilinear_sampler :7-17).  Level 1 (cv averaged over pairs of j) is computed on the fly, never stored.
 *   cv    (B, h, w, w) dtype cv_dtype, volume rows cv_pitch elements apart (0 = w);  disp (B, h, w) fp32
 *   corr1, corr2  fp32 (or fp16 when out_dtype = S2M2_F16) with element (b,y,i,k) at
 *       base + ((b*h + y)*w + i)*pix_stride + k*tap_stride      k = 0..2r  <->  dx = k - r
 *   (planar (B,2r+1,h,w) of the reference: pix_stride=1, tap_stride=h*w with base offset b*(2r+1)*h*w handled by
 *    batch_stride; channels-last: pix_stride=2r+1 (or larger), tap_stride=1)
 *   The reference's pixel -> normalised -> pixel fp32 coordinate round trip (grid_sample, align_corners=True,
 *   zeros padding) is reproduced on both axes.
 */
int s2m2_cv_lookup(const void* cv, const float* disp, void* corr1, void* corr2,
                   int B, int h, int w, int radius, int cv_dtype, int out_dtype,
                   long long batch_stride, long long pix_stride, long long tap_stride, int cv_pitch, void* stream);

/*
 * [A2,A3,A7,A8,A11,A12,A14,A15] implicit-GEMM convolution / linear layer with fused concatenation and epilogues.
 *   Replaces the stride-1 nn.Conv2d / nn.ConvTranspose2d / nn.Linear call sites outside the CNN backbone
 *   (refinenet.py:14-20,47-57,87-122; attentions.py:24-28,71-74,239-241,269-275; feature_fusion.py:15-21;
 *   stacked_MRT.py:22-34; unet.py:25-37; submodules.py:104-108,127-135; s2m2.py:65-67) and the elementwise kernels
 *   PyTorch launches around them: torch.cat of the inputs, bias, GELU/ReLU/sigmoid/tanh, residual add, the ConvGRU gate
 *   arithmetic (refinenet.py:24-34), the FeatureFusion gate mix (feature_fusion.py:24-31).
 *
 *   out[n,y,x,co] = epi( out_scale * act( bias[co] + sum_{ky,kx,ci} in[n, y*stride+ky-KH/2, x*stride+kx-KW/2, ci] * weight[co,ky,kx,ci] ) )
 *   in  = channel concatenation of src[0..nsrc) (NHWC, src_c[s] channels each, pixel stride src_stride[s]; zero padding)
 *   weight  packed (Cout, KH*KW, Cin), Cin = sum(src_c), same dtype as the activations; bias fp32 (Cout) or NULL
 *   out NHWC with pixel stride out_stride (write into a channel slice of a wider tensor by offsetting `out`)
 *   every channel count / stride is a multiple of 8 (callers zero-pad: weights of padded channels are zero)
 *   epi: ADD  v + aux0 | MUL  v * aux0 | GRU  (1-aux0)*aux1 + aux0*v  (aux0 = z, aux1 = h, v = q)
 *        GATEMIX  g = clamp(v, .01, .99); g*aux0 + (1-g)*aux1          (aux tensors NHWC at the output pixel/channel)
 *        DUALMIX  two 1x1 layers on the same rows in one launch (the two heads of FeatureFusion, feature_fusion.py:15-31):
 *                 weight rows = [W1 (Cout x ksplit) | W2 (Cout x (Cin - ksplit))] along K, act = SIGMOID;
 *                 g = clamp(sigmoid(W1.in[:ksplit] + bias), .01, .99);  out = (W2.in[ksplit:] + bias2) + g*aux0 + (1-g)*aux1
 *   shuffle2 = C' > 0: the conv is the GEMM of a ConvTranspose2d(kernel 2, stride 2): KH = KW = 1, Cout = 4*C' ordered
 *        (dy, dx, c'), result stored to (N, 2H, 2W, C') with pixel stride out_stride.
 *   tile: 0 = automatic, 1..4 force a block tile (128x128, 64x64, 128x32, 128x64) -- tests and tuning only.
 */
enum { S2M2_ACT_NONE = 0, S2M2_ACT_GELU = 1, S2M2_ACT_RELU = 2, S2M2_ACT_SIGMOID = 3, S2M2_ACT_TANH = 4 };
enum { S2M2_EPI_NONE = 0, S2M2_EPI_ADD = 1, S2M2_EPI_MUL = 2, S2M2_EPI_GRU = 3, S2M2_EPI_GATEMIX = 4, S2M2_EPI_DUALMIX = 5 };
typedef struct s2m2_conv_desc {
    const void* src[4];
    int src_c[4];
    int src_stride[4];
    int nsrc;
    const void* weight;
    const float* bias;
    void* out;
    int out_stride;
    int N, H, W, KH, KW, Cout;
    int act, epi;
    const void* aux0;
    const void* aux1;
    int aux0_stride, aux1_stride;
    float out_scale;
    int shuffle2;
    int tile;
    int dtype;
    int korder;             /* K order of the packed weight: 0 = (Cout, KH, KW, Cin); 1 = (Cout, KH, Cin/CH, KW, CH) with CH = 64 bytes of
                               channels (32 fp16 / 16 fp32; Cin % CH == 0): horizontal taps become consecutive K tiles -> L1 reuse;
                               2 = fp16 MFMA fragment stream [Cout/32][ceil(Cin/CK)][KH*KW][CK/16][64 lanes][8] (lane l: cout 32t + l%32,
                               channel CK c + 16s + 8(l/32) + e; zero beyond Cin; CK = s2m2_conv_frag_chunk(Cout, Cin)) for stride-1
                               3x3 / 3x1 / 1x3 layers with Cout % 128 == 0, or Cout % 192 == 0 and Cin % 192 == 0:
                               the weights go from L2 straight into MFMA operands, only the input patch is staged in LDS */
    int stride;             /* 1 or 2: out[y,x] is centred on in[y*stride, x*stride]; output (N, ceil(H/stride), ceil(W/stride), Cout) */
    const float* ln_wsum;   /* non-NULL: the layer is LayerNorm(Cin, elementwise_affine=False, eps=ln_eps) followed by this 1x1 layer
                               (the pre-norm of attentions.py:117,148,182,213,243 feeding its Linear): `in` holds the RAW rows, the kernel
                               computes W.((x-mean)*rstd)+b as rstd*(W.x - mean*ln_wsum)+b with the row statistics taken inside the GEMM;
                               ln_wsum[co] = sum_k weight[co,k] (fp32, Cout entries, summed from the packed weight).
                               Needs KH = KW = 1, stride 1, no shuffle2, Cin with no padding channels, act NONE or GELU. */
    float ln_eps;
    int ksplit;             /* S2M2_EPI_DUALMIX: first K index (channel of `in`) of the second layer; multiple of 64 */
    const float* bias2;     /* S2M2_EPI_DUALMIX: bias of the second layer (fp32, Cout) or NULL */
    int pool2;              /* 1: the layer is nn.AvgPool2d(2) followed by this 1x1 layer (down_conv of Unet / MRT, reference unet.py:24-29,
                               stacked_MRT.py:21-26): a GEMM row is the mean of input pixels (2y, 2x) .. (2y+1, 2x+1), formed and rounded
                               to the I/O dtype exactly as s2m2_resample2x mode 0 does; output (N, H/2, W/2, Cout).  Needs KH = KW = 1,
                               stride 1, korder 0, no shuffle2 / ln_wsum / DUALMIX. */
    int epi_cout0;          /* > 0 (korder 2, one-operand epilogues ADD / MUL, a multiple of 128): the epilogue applies to couts >= epi_cout0
                               only, couts below it are stored after bias + activation -- two layers that read the same input stacked along
                               Cout with different epilogues in ONE launch (ConvGRU: z = sigmoid(convz(hx)) | r*h = sigmoid(convr(hx)) * h,
                               refinenet.py:24-29).  aux0 then has Cout - epi_cout0 channels: aux0[pixel * aux0_stride + (cout - epi_cout0)]
                               (ABI 500: the tensor's own base pointer; up to ABI 400 the caller passed it shifted by -epi_cout0 elements). */
} s2m2_conv_desc;
int s2m2_conv2d(const s2m2_conv_desc* desc, void* stream);

/*
 * K9 -- a chain of up to three 1x1 layers (nn.Linear / 1x1 nn.Conv2d, all C -> C) on token rows in ONE launch; the row tile
 *   stays in LDS between the layers.  Replaces, per transformer block, the attention output projection + residual and the FFN
 *   (reference attentions.py:311-321 GlobalAttnBlock.forward, :347-355 BasicAttnBlock.forward:  z = z + proj(o);
 *   z = z + ffn(norm(z)) with ffn = Linear-GELU-Linear, attentions.py:239-241), and the 1x1 branch of ConvBlock2D
 *   (attentions.py:269-275: Conv1x1-ReLU-Conv1x1).
 *
 *   t_0 = x rows;   y_s = act_s( W_s . (ln_wsum[s] ? LayerNorm(t_s) : t_s) + b_s );   if (s == res_stage) y_s += res rows;
 *   if (carry && s == 2) y_2 += y_0;   t_{s+1} = y_s;   out = y_{nstage-1}
 *   weight[s] packed (C, C) like s2m2_conv2d's 1x1 weight, bias fp32 (C) or NULL, ln_wsum[s] fp32 (C) row sums of weight[s]
 *   (non-NULL: LayerNorm without affine, eps ln_eps, folded in as in s2m2_conv2d); act NONE / GELU / RELU.
 *   Every y_s is rounded to the I/O dtype before it is used again, exactly as separate launches would store it.
 *   C: 128 / 256 / 384 / 512 (fp16), 128 / 256 (fp32): ask s2m2_mlp_chain_supported; row strides in elements, multiples of 8.
 */
typedef struct s2m2_chain_desc {
    const void* x;
    const void* res;
    void* out;
    long long x_stride, res_stride, out_stride, rows;
    int C, nstage;
    const void* weight[3];
    const float* bias[3];
    const float* ln_wsum[3];
    int act[3];
    int res_stage;          /* -1: no residual rows */
    int carry;
    float ln_eps;
    int dtype;
    /* optional second output of the last stage: ln_out rows = LayerNorm(out rows) * ln_gamma + ln_beta over the C channels (fp32
       statistics of the rounded `out` rows, biased variance, eps ln_out_eps, result rounded to the I/O dtype).  Folds DispInit's
       layer_norm (submodules.py:165,216) into the launch that produces feature_tr_4x; consumed by s2m2_corr.  NULL: none.
       Needs C * sizeof(dtype) / 16 in {16, 32, 64} (fp16: C = 128 / 256 / 512; fp32: C = 128 / 256). */
    void* ln_out;
    long long ln_out_stride;
    const float* ln_gamma;
    const float* ln_beta;
    float ln_out_eps;
    /* fan-out stages, nfan = 0: none.  nfan further C -> C layers that ALL read the chain's `out` rows (while they are still in LDS) and
       write fan_out[:, f*C:(f+1)*C] = W_f . (fan_ln_wsum ? LayerNorm(out rows) : out rows) + b_f  -- the Q | K | V projection of the attention
       block that follows (reference attentions.py:24-28,71-74 behind the pre-norm of :117,148), fused into the launch that produces its
       input: fan_weight packed (nfan*C, C), fan_bias fp32 (nfan*C) or NULL, fan_ln_wsum fp32 (nfan*C) row sums (eps ln_eps) or NULL. */
    const void* fan_weight;
    const float* fan_bias;
    const float* fan_ln_wsum;
    void* fan_out;
    long long fan_out_stride;
    int nfan;
    /* placement hint, 0 = none: the rows are `rows / (8 * xcd_group_rows)` images of 8 groups of xcd_group_rows consecutive rows each;
       group g of every image is processed on XCD g (blocks are dealt to XCDs round robin by the hardware), so that a consumer which
       places its work the same way -- s2m2_corr: image row y on XCD y / (h / 8) -- reads these rows from the L2 that holds them. */
    long long xcd_group_rows;
    /* 1: weight[s] and fan_weight are in MFMA-FRAGMENT order instead of row-major -- the same C x C (nfan*C x C) fp16 values, with the
       16-byte piece (cout o, channels 8q .. 8q+7) at 16-byte slot ((o/32) * (C/16) + q/2) * 64 + (q%2) * 32 + o%32 (per fan-out layer for
       fan_weight) -- and the launch runs the DIRECT form: a wave's weight fragments go from global memory straight into its MFMA operand
       registers, one whole stage ahead, no weight tile in LDS and no block barrier inside a stage.  For SHORT row counts (the 1/32 .. 1/8
       pyramid levels: a block lives for the latency of its weight stream, not for its arithmetic).  fp16, C = 128 / 192 / 256 / 384 / 512: ask
       s2m2_mlp_chain_frag_supported.  Same arithmetic and rounding points as the row-major form.  With nstage = 0 and nfan = 1 .. 4 the
       fan-out layers alone run in this form (any row count; nstage = 0 exists in this form only: ask s2m2_mlp_fan_supported). */
    int weight_frag;
    /* > 0 (with weight_frag): x is an image tensor (N, pool_h, pool_w, C) with pixel stride x_stride, and row m = (n, yo, xo) of the
       (pool_h/2, pool_w/2) grid is the mean of its four pixels (2yo, 2xo) .. (2yo+1, 2xo+1): nn.AvgPool2d(2) in front of the 1x1 layer(s)
       (the down_conv of Unet / MRT, reference unet.py:24-29, stacked_MRT.py:21-26) folded into the tile load, rounded to fp16 like the
       stand-alone pooling launch; rows = N * (pool_h/2) * (pool_w/2).  With chain stages (nstage >= 1, ABI 600) or fan-out only; residual,
       carry, ln_out and xcd_group_rows are not combined with it. */
    int pool_h, pool_w;
} s2m2_chain_desc;
int s2m2_mlp_chain_supported(int C, int dtype);
int s2m2_mlp_chain_frag_supported(int C, int dtype);
/* nstage = 0 with nfan > 0 ("fan-out only": the nfan layers read the x rows themselves -- a Q | K | V projection, or a pooled down_conv, as
 * ONE pass over the rows): 1 where the library has that form (the direct form, weight_frag = 1: fp16, C = 128 / 256, nfan 1..4), else 0 */
int s2m2_mlp_fan_supported(int C, int nfan, int dtype);
int s2m2_mlp_chain(const s2m2_chain_desc* desc, void* stream);

/*
 * K13 -- one whole 1-D (epipolar) attention step on token rows in ONE launch (ABI 600): pre-LayerNorm -> Q | K | V projections -> softmax
 *   attention along the image row -> output projection + residual -> pre-LayerNorm -> FFN (Linear - GELU - Linear) + residual.  Replaces, per
 *   step, the launch triple s2m2_mlp_chain(fan-out Q|K|V) / s2m2_attention / s2m2_mlp_chain of the reference's
 *   CrossAttnBlock1D + FFN and SelfAttnBlock1D + FFN (attentions.py:131-161, :99-128, :229-250; BasicAttnBlock.forward :347-355):
 *     z' = z + proj( softmax( (LN(z) Wq^T)(LN(s) Wk^T)^T / sqrt(d) ) (LN(s) Wv^T + bv) ),   out = z' + W2 GELU(W0 LN(z') + b0) + b2
 *   with z = a token row (image n, line y) and s = the same line of image (n + nimg/2) % nimg (cross = 1: left <-> right, shared weights, both
 *   directions in the launch) or z itself (cross = 0).  One block per token row; Q, K, V, the attention output and the FFN hidden tensor never
 *   leave the CU (K / V of the source row in LDS, everything else in registers).  Rounding points = those of the separate launches.
 *     x, out      (nimg, h, w, 128) fp16, channels contiguous; x_stride / out_stride = elements between tokens (multiples of 4); out != x
 *     weights     the six layers q, k, v, proj, ffn.0, ffn.2 back to back, 128 x 128 fp16 each (row = output channel) in the "row_attn
 *                 packing", 16-byte aligned, 32 KB per layer:
 *                 (1) columns: per 16 input channels the four quads of 4 channels in the order (0, 2, 1, 3) -- column 16g + 8a + 4b + e of the
 *                 plain packing (s2m2_conv2d's 1x1 weight) holds input channel 16g + 8b + 4a + e (a, b in {0, 1}, e < 4; an involution): the
 *                 k-slots of a 16-byte MFMA fragment then follow the accumulator layout of the layer before it;
 *                 (2) unit-major: the 16-byte unit u (columns 8u .. 8u+7 of (1)) of row r at element (u * 128 + r) * 8 of the layer -- the
 *                 kernel copies a layer into LDS with a linear LDS-DMA and reads conflict-free fragments at immediate offsets
 *     vectors     twelve fp32 vectors of 128 back to back, 16-byte aligned (absent biases as zeros): 0 bias q, 1 row sums of q, 2 bias k,
 *                 3 row sums of k, 4 bias v, 5 row sums of v, 6 bias proj, 7 bias ffn.0, 8 row sums of ffn.0, 9 bias ffn.2, 10 ln_out gamma,
 *                 11 ln_out beta.  Row sums (of the plain fp16 weight, in fp32) fold the LayerNorm without affine (eps ln_eps) in front
 *                 of q, k, v, ffn.0 as in s2m2_conv2d (ln_wsum); 10 / 11 are used only with ln_out
 *     ln_out      optional second output: LayerNorm(out) * gamma + beta (eps ln_out_eps) -- DispInit's layer_norm (submodules.py:165,216)
 *                 when this step writes feature_tr_4x, as s2m2_chain_desc.ln_out
 *     xcd_hint    1 (h % 8 == 0): line y of every image runs on XCD y / (h / 8) -- where s2m2_cost_volume reads the rows (see
 *                 s2m2_chain_desc.xcd_group_rows) and where the partner row's block runs
 *   fp16, C = 128, heads 1 or 2 (head dim 128 / 64), 8 <= w <= 320: ask s2m2_row_attn_supported.
 */
typedef struct s2m2_rowattn_desc {
    const void* x;
    long long x_stride;
    void* out;
    long long out_stride;
    int nimg, h, w, C;
    int heads;
    int cross;
    const void* weights;
    const float* vectors;
    float ln_eps;
    void* ln_out;
    long long ln_out_stride;
    float ln_out_eps;
    int xcd_hint;
    int dtype;
} s2m2_rowattn_desc;
int s2m2_row_attn_supported(int C, int heads, int w, int dtype);
int s2m2_row_attn(const s2m2_rowattn_desc* desc, void* stream);

/*
 * K14 -- a whole ConvBlock2D (attentions.py:255-281) in ONE launch, for the coarse pyramid levels (ABI 600):
 *     out = convs.2( GELU( convs.0(x) ) ) + convs_1x.2( ReLU( convs_1x.0(x) ) )          convs.* 3x3 (padding 1), convs_1x.* 1x1, all C -> C
 *   Replaces the launch triple s2m2_mlp_chain (1x1 branch) / s2m2_conv2d (korder 2) / s2m2_conv2d (korder 2, EPI_ADD) with the same arithmetic
 *   in the same order (bit-identical): a block owns a patch of output pixels and all C channels, recomputes the one-pixel ring of the first
 *   3x3 layer and keeps the GELU tensor in LDS.  Meant for grids whose launches are latency chains (1/8 ... 1/32 resolution).
 *     x, out     (N, H, W, C) fp16, channels contiguous, pixel strides in elements (x: multiple of 8, out: multiple of 4); out != x
 *     w_conv0/2  the 3x3 layers as s2m2_conv_desc.weight with korder = 2 (s2m2_pack_frag S2M2_PACK_CONV_FRAG, 128-channel chunks)
 *     w_1x0/2    the 1x1 layers as s2m2_chain_desc.weight with weight_frag = 1 (S2M2_PACK_ROWS)
 *     b_*        fp32 (C) or NULL
 *     patch_rows 0 = the library's choice; 2 / 4 force the patch height (C = 128; C = 256 runs 2-row patches)
 *   fp16, C = 128 / 256: ask s2m2_conv_block_supported.
 */
typedef struct s2m2_convblock_desc {
    const void* x;
    long long x_stride;
    void* out;
    long long out_stride;
    int N, H, W, C;
    const void* w_conv0;
    const void* w_conv2;
    const void* w_1x0;
    const void* w_1x2;
    const float* b_conv0;
    const float* b_conv2;
    const float* b_1x0;
    const float* b_1x2;
    int patch_rows;
    int dtype;
} s2m2_convblock_desc;
int s2m2_conv_block_supported(int C, int H, int W, int dtype);
int s2m2_conv_block(const s2m2_convblock_desc* desc, void* stream);

/*
 * K11 -- a 1x1 layer with any channel counts in the direct style (fp16; round 4): Conv2d(kernel 1) / Linear / ConvTranspose2d(2, stride 2) on
 *   up to four channel-concatenated sources (reference: LocalRefiner's corr_feat / conf_occ_feat / disp_corr_ctx_cat 1x1 layers,
 *   refinenet.py:87-106,138-146; the up_conv 1x1 layers of Unet / MRT on the coarse grid, unet.py:32-37, stacked_MRT.py:29-34; the
 *   ConvTranspose heads of the upsampling masks, submodules.py:104-113,131-144), bias, NONE / GELU / RELU.
 *     out[m, :] = act(W . cat_i(src_i[m, :]) + bias)      rows m of `rows` tokens; src_i: src_c[i] channels (multiples of 8), row stride
 *   src_stride[i]; K = sum src_c, K rounded up to 16 in {32, 48, 64, 96, 128, 192, 256, 384}; Cout a multiple of 8 with 1, 2, 3, 4, 6 or 8
 *   32-cout tiles: ask s2m2_pw_direct_supported(K, Cout, dtype).
 *   weight_frag: the (Cout, K) weight zero-padded to (32 * tiles, 16 * steps) in MFMA-fragment order -- 16-byte slot (t * steps + s) * 64 + l
 *   holds row 32 t + l % 32, columns 16 s + 8 (l / 32) .. + 7 (pack.pw_frag); bias fp32 (Cout) or NULL.
 *   shuffle2 > 0: the ConvTranspose2d(2, s 2) store of s2m2_conv_desc (Cout = 4 * shuffle2, rows = N * Ho * Wo input pixels).
 */
typedef struct s2m2_pw_desc {
    const void* src[4];
    int src_c[4];
    long long src_stride[4];
    int nsrc;
    long long rows;
    const void* weight_frag;
    const float* bias;
    void* out;
    long long out_stride;
    int Cout;
    int act;
    int shuffle2, Ho, Wo;
    int dtype;
} s2m2_pw_desc;
int s2m2_pw_direct_supported(int K, int Cout, int dtype);
int s2m2_pw_direct(const s2m2_pw_desc* desc, void* stream);

/*
 * K12 -- spatial convolutions on NARROW inputs in the direct style (fp16; round 4): Conv2d / stride-1 ConvTranspose2d (as a convolution with
 *   the flipped kernel) whose input has exactly 8 or 16 channels -- one or two 16-byte pieces per pixel (reference: UpsampleMask1x
 *   conv_disp.0 | conv_rgb.0, submodules.py:124-129,139-141; LocalRefiner disp_feat.0 | conf_occ_feat.0, refinenet.py:93-101,141-142;
 *   CNNEncoder conv1_down.0, submodules.py:69-71).  Shapes: 3x3 stride 1 on 8 channels (any Cout % 8 == 0), 5x5 stride 2 on 16 channels
 *   (an even number of 32-cout tiles), and the second form below: ask s2m2_conv_narrow_supported (Cin = all input channels).  Padding K / 2, Ho = ceil(H / stride) as s2m2_conv2d.
 *     out[n, y, x, :] = act(sum_taps W[:, tap, :] . x[n, y*s - K/2 + ky, x*s - K/2 + kx, :] + bias)        act: NONE / GELU / RELU
 *   x: (N, H, W, Cin) with pixel stride x_stride (elements, multiple of 8); out: (N, Ho, Wo, Cout), pixel stride out_stride.
 *   weight_frag: the (Cout, KH*KW*Cin) matrix of s2m2_conv2d's K order 0 (K = (tap, channel)) zero-padded to (32 * tiles, 16 * steps) in
 *   MFMA-fragment order, as for K11 (pack.pw_frag); bias fp32 (Cout) or NULL.
 *   Second form, same entry point -- 3x3 stride 1 with FEW OUTPUT channels on wider inputs (UpsampleMask1x conv_concat.0, submodules.py:
 *   133-137,143; LocalRefiner disp_feat.2 and disp_update.2 | conf_occ_update.2, refinenet.py:93-96,108-118; GlobalRefiner out_feat,
 *   refinenet.py:61-66): Cin = 48 (Cout <= 64), 96 (Cout <= 96), 128 / 256 (Cout <= 32), from one or two channel-concatenated sources
 *   (x, x1).  Cin = 128 / 256: the K columns of the weight matrix chunk-major, K = (chunk of 64 channels, tap, channel) (pack.narrow_frag).
 */
typedef struct s2m2_narrow_desc {
    const void* x;
    long long x_stride;
    int N, H, W, Cin;                  /* Cin: ALL input channels (x holds the first Cin - Cin1 of them) */
    const void* x1;                    /* second source: channels [Cin - Cin1, Cin), pixel stride x1_stride; Cin1 = 0: none */
    long long x1_stride;
    int Cin1;
    const void* weight_frag;
    const float* bias;
    void* out;
    long long out_stride;
    int Cout, KH, KW, stride;
    int act;
    int dtype;
    /* fused 1x1 head (Cin = 48 form only; UpsampleMask1x conv_concat.0 -> ReLU -> conv_concat.2, submodules.py:133-137,143-144):
       head_cout > 0 (act NONE or RELU): out holds  W2 . act(conv3x3(x) + bias) + head_bias  with head_cout (multiple of 8, <= 32) channels instead of the Cout
       channels of the 3x3 layer, which are rounded to fp16 as a store would round them and never leave the registers.  head_frag: the
       (head_cout, Cout) matrix W2, rows zero-padded to 32, K columns zero-padded to 32-channel tiles and ordered to match the accumulator layout
       of the 3x3 layer: k16 step (j, p), j = channel / 32, p < 2; 16-byte slot ((j * 2 + p) * 64 + l) holds row l % 32 and the channels
       32 j + 8 (2 p + q) + 4 (l / 32) + e for q = 0, 1 and e = 0 .. 3, in that order (pack.head_frag). */
    const void* head_frag;
    const float* head_bias;
    int head_cout;
} s2m2_narrow_desc;
int s2m2_conv_narrow_supported(int KH, int KW, int stride, int Cin, int Cout, int dtype);
int s2m2_conv_narrow(const s2m2_narrow_desc* desc, void* stream);

/*
 * K10 -- FeatureFusion with 1x1 kernels in ONE launch (reference feature_fusion.py:4-33 with kernel_size = 1: every fusion of
 *   Unet / MRT, unet.py:39-41, stacked_MRT.py:36-41), the 3C-wide hidden tensor never leaves the CU:
 *     h = GELU(w1 . cat(z0, z1) + b1);  g = clamp(sigmoid(Wg . h[:C] + bg), .01, .99);  out = (Wf . h[C:] + bf) + g*z0 + (1-g)*z1
 *   z0, z1, out: `rows` token rows of C channels (row strides in elements, multiples of 8), dtype `dtype`;
 *   w1 packed (3C, 2C): rows [0,C) = feature_gate.0, rows [C,3C) = feature_fusion.0;  w2 packed (C, 3C) = [Wg (C,C) | Wf (C,2C)]
 *   along K;  b1 (3C), bg (C), bf (C) fp32.  C = 128 or 256: ask s2m2_feature_fusion_supported.
 *   z1_coarse_h / z1_coarse_w > 0: z1 is the COARSE tensor (N, z1_coarse_h, z1_coarse_w, C) and the kernel reads it through
 *   nn.Upsample(scale_factor=2, mode='bilinear', align_corners=False) (unet.py:32-37: the up_conv feeding every decoder fusion);
 *   rows = N * 2*z1_coarse_h * 2*z1_coarse_w in raster order.  0 / 0: z1 has one row per output row.
 */
int s2m2_feature_fusion_supported(int C, int dtype);
int s2m2_feature_fusion(const void* z0, const void* z1, void* out, long long z0_stride, long long z1_stride, long long out_stride,
                        long long rows, int C, const void* w1, const float* b1, const void* w2, const float* bg, const float* bf,
                        int z1_coarse_h, int z1_coarse_w, int dtype, void* stream);
/*
 * K10, direct form (fp16, C = 128 / 256: ask s2m2_feature_fusion_frag_supported) for SHORT row counts: the same block, with the weights of
 *   both layers as ONE stream of 1 KB MFMA fragments per 32-cout tile, in the order the kernel consumes them, read from global memory
 *   straight into the operand registers (no weight tile in LDS, 6 block barriers per block instead of one per 64-byte K chunk).
 *   w_stream: 9*C*C fp16 values; fragment j of cout tile t at 16-byte slot (t * 9C/16 + j) * 64 + lane, lane l holding 8 consecutive K
 *   elements (k16 step, half l/32) of row 32t + l%32 of:  for slice s = 0, 1, 2:  w1 rows [sC, sC + C) (k16 steps 0 .. 2C/16),  then
 *   w2 = [Wg | Wf] columns [sC, sC + C) (steps 0 .. C/16)  -- w1, w2 as in s2m2_feature_fusion.  Same arithmetic and rounding points.
 */
int s2m2_feature_fusion_frag_supported(int C, int dtype);
int s2m2_feature_fusion_frag(const void* z0, const void* z1, void* out, long long z0_stride, long long z1_stride, long long out_stride,
                             long long rows, int C, const void* w_stream, const float* b1, const float* bg, const float* bf,
                             int z1_coarse_h, int z1_coarse_w, int dtype, void* stream);

/*
 * [A2,A3] pre-norm LayerNorm without affine over the channel axis (attentions.py:117,148,182,213,243; eps 1e-5, biased var).
 *   x, y: `rows` token rows of C channels, row strides x_stride / y_stride elements (multiples of 8); fp32 arithmetic.
 */
int s2m2_layernorm(const void* x, void* y, long long rows, int C, long long x_stride, long long y_stride, int dtype, void* stream);

/*
 * [A1] nn.GroupNorm(G, C) with affine on an NHWC activation (CNNEncoder, submodules.py:80,90).  x, y: (N, HW, C);
 *   gamma, beta fp32 (C); workspace >= s2m2_groupnorm_workspace_bytes(N, G) bytes (zeroed inside, on the stream).
 *   Statistics are accumulated in fp64, the normalisation runs in fp32.
 */
size_t s2m2_groupnorm_workspace_bytes(int N, int G);
int s2m2_groupnorm_nhwc(const void* x, void* y, const float* gamma, const float* beta, void* workspace, int N, long long HW,
                        int C, int G, float eps, int dtype, void* stream);

/*
 * [A14,A15 tails] convex upsampling (S2M2.upsample4x / upsample1x, s2m2.py:101-133; custom_unfold utils.py:9-20):
 *   out[m][b,Y,X] = scale[m] * sum_{n<9} softmax_n(logits[b,Y,X,0:9]) * x[m][b, clamp(Y/factor + n/3 - 1), clamp(X/factor + n%3 - 1)]
 *   x[m]   (B, hs, ws) fp32, m < nmaps <= 3 (host array of device pointers);  out[m] (B, hs*factor, ws*factor) fp32
 *   logits NHWC at the OUTPUT resolution, 9 used channels, rows padded to logit_stride >= 16 elements, dtype `dtype`;
 *   logit_up2 = 1 (output_upsample, s2m2.py:123-127): logits are (B, hs, ws, .) and bilinearly upsampled x2
 *   (align_corners=False, rounded to `dtype`) on the fly; factor must be 2.
 *   chan_out != NULL: map 0 is also stored, in `dtype`, at chan_out[pixel * chan_stride] (the disparity input channel of
 *   UpsampleMask1x, submodules.py:137, written straight into the 8-channel image tensor).
 */
int s2m2_convex_upsample(const float* const* x, float* const* out, const float* scale, int nmaps, const void* logits,
                         int logit_stride, int B, int hs, int ws, int factor, int logit_up2, void* chan_out,
                         long long chan_stride, int dtype, void* stream);

/*
 * [A2,A3] multi-head attention softmax(Q K^T * scale) V, flash style (no score matrix in memory).  Replaces
 *   F.scaled_dot_product_attention and the explicit attention + positional-encoding einsums (attentions.py:42-50, :83-91).
 *   q, k, v, out: token rows; element (batch b, token n, head hd, e) at base + (b*N + n)*stride + hd*D + e  (so the fused
 *   QKV projection is read in place and the output is (tokens, heads*D)).  D multiple of 8, <= 256 (<= 128 with PE).
 *   swap_halves = 1: keys/values of batch b come from batch (b + nb/2) % nb (symmetric cross attention, CrossAttn).
 *   pe_x != NULL selects SelfAttn(use_pe=True): tokens form a grid_h x grid_w grid (row major), pe_x (2*grid_w-1, 16) and
 *   pe_y (2*grid_h-1, 16) fp32 are the separable sinc tables of get_pe (utils.py:32-60) indexed by xq-xk+grid_w-1 /
 *   yq-yk+grid_h-1, and pe_out (tokens, heads*32) receives sum_k P[q,k] * 0.5*[pe_x[..], pe_y[..]]  (the input of pe_proj).
 */
int s2m2_attention(const void* q, const void* k, const void* v, void* out, long long q_stride, long long k_stride,
                   long long v_stride, long long out_stride, int nb, int heads, int Nq, int Nk, int D, float scale,
                   int swap_halves, const float* pe_x, const float* pe_y, void* pe_out, long long pe_stride,
                   int grid_w, int grid_h, int dtype, void* stream);
/*
 * Plans the launch s2m2_attention would make for (nb, heads, N = Nq = Nk, D, dtype) -- head-dim instantiation, and for the PE variant
 *   (grid_w, grid_h > 0; 0, 0 = no positional encoding) the marginal-bin tiles, waves per block and the 160 KB LDS budget -- without
 *   launching.  1 = supported, 0 = not (s2m2_last_error names the limit).  Lets the host reject a geometry BEFORE the first launch of a
 *   forward (token grids above 96 x 96 cells = images above 3072 px per side have no PE instantiation).
 */
int s2m2_attention_supported(int nb, int heads, int N, int D, int grid_w, int grid_h, int dtype);

/*
 * [A2,A3] 2x resampling of an NHWC activation: mode 0 = nn.AvgPool2d(2) (unet.py:25-30, stacked_MRT.py:22-27), mode 1 =
 *   bilinear x2 with align_corners=False (unet.py:32-37, stacked_MRT.py:29-34).  x (N,H,W,C) -> y (N,H/2,W/2,C) or (N,2H,2W,C),
 *   pixel strides x_stride / y_stride elements; fp32 arithmetic.
 */
int s2m2_resample2x(const void* x, void* y, int N, int H, int W, int C, long long x_stride, long long y_stride, int mode,
                    int dtype, void* stream);

/*
 * [A0,A7,A11,A13] fused per-pixel stages between the big kernels (each is a chain of separate elementwise kernels in PyTorch):
 *   s2m2_image_prep     normalize_img (s2m2.py:80-89) + left/right concat (:143) + NHWC packing: img0, img1 (B,3,H,W) planar,
 *                       img_dtype S2M2_F32 / S2M2_F16 / 2 (uint8), values in [0,255] -> x8 (2B,H,W,8) with channels 1..3 =
 *                       (v/255 - 0.5)*2 and channels 0, 4..7 = 0 (channel 0 later carries the upsampled disparity, A15)
 *   s2m2_refine_prep    mode 0 (GlobalRefiner, refinenet.py:63-68): small8[...,0] = disp/100*mask, [...,1] = logit(mask*conf, 0.1),
 *                       mask = conf > 0.2;  mode 1 (LocalRefiner, :134-141): disp/100, logit(conf, 0.01), logit(occ, 0.01)
 *   s2m2_global_update  out = mask*disp + (1-mask)*upd*100 [clamped at 0]  (refinenet.py:70-71, s2m2.py:160-161); upd = channel 0
 *                       of an NHWC tensor with pixel stride upd_stride
 *   s2m2_refine_update  in place: disp += dco[0]; conf = sigmoid(dco[8] + logit(conf, .01)); occ likewise with dco[9]
 *                       (refinenet.py:149-151); then clamp disp at 0 if use_positivity and occ *= (x - disp >= 0) (s2m2.py:177-180)
 *   s2m2_refine_update_to  the same out of place (outputs may alias the inputs); small8_next != NULL also receives the mode-1
 *                       side input of the next refinement iteration (= s2m2_refine_prep of the values just written, bit for bit)
 *   s2m2_tanh           y = tanh(x) on n elements (hidden = tanh(ctx), s2m2.py:166)
 *   s2m2_stem_mlp       the two 1x1 layers at the head of CNNEncoder on full-resolution pixels (submodules.py:68-71: conv0 =
 *                       Conv2d(3,16,1) - GELU - Conv2d(16,16,1)): x8 (npix,8) -> out (npix,16) = W1.gelu(W0.x + b0) + b1;
 *                       w0 (16,8), w1 (16,16), biases (16): fp32, weights already rounded to the activation dtype
 *   disp, conf, occ, out: (B,h,w) fp32.
 */
int s2m2_image_prep(const void* img0, const void* img1, void* x8, int B, int H, int W, int img_dtype, int dtype, void* stream);
int s2m2_refine_prep(const float* disp, const float* conf, const float* occ, void* small8, long long npix, int mode, int dtype, void* stream);
int s2m2_global_update(const void* upd, int upd_stride, const float* disp, const float* conf, float* out, long long npix, int clamp0,
                       int dtype, void* stream);
int s2m2_refine_update(const void* dco, int dco_stride, float* disp, float* conf, float* occ, long long npix, int w, int use_positivity,
                       int dtype, void* stream);
int s2m2_refine_update_to(const void* dco, int dco_stride, const float* disp, const float* conf, const float* occ, float* disp_out,
                          float* conf_out, float* occ_out, void* small8_next, long long npix, int w, int use_positivity, int dtype,
                          void* stream);
int s2m2_tanh(const void* x, void* y, long long n, int dtype, void* stream);
int s2m2_stem_mlp(const void* x8, const float* w0, const float* b0, const float* w1, const float* b1, void* out, long long npix, int dtype,
                  void* stream);

/*
 * [8f-1] image_pad of the reference driver (src/s2m2/core/utils/image_utils.py:27-71), on the device: (B,C,H,W) planar image
 *   (img_dtype S2M2_F32 / S2M2_F16 / 2 = uint8) -> out (B,C,Hn,Wn) fp32, Hn/Wn = H/W rounded up to multiples of `factor`, the
 *   original image centred (offset (Hn-H)/2, (Wn-W)/2); the border = bilinear (align_corners=False) upsampling of the adaptive
 *   average pooling of the zero-padded image to (H/factor, W/factor).  pooled: scratch (B,C,H/factor,W/factor) fp32.
 */
int s2m2_image_pad(const void* img, float* pooled, float* out, int B, int C, int H, int W, int factor, int img_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* S2M2_HIP_H */
