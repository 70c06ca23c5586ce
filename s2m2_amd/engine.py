"""Inference engine behind ``S2M2.forward`` on MI355X: every stage of the reference forward (s2m2.py:136-197) runs as
hand-written gfx950 kernels through the C ABI of ``libs2m2_hip.so`` (:mod:`s2m2_amd.hip`).  PyTorch is plumbing only: it owns
device memory and the stream, packs the weights once, and prepares a few tiny (B,h,w,8) side inputs of the refiners.

Activations are NHWC ``(N, H, W, C)`` tensors (channels contiguous, every channel count a multiple of 8), which is at the same
time the token layout of the attention blocks.  Stage map (reference file:line -> kernel):

* CNNEncoder (submodules.py:63-93), every Conv2d / ConvTranspose2d / Linear of Unet, MRT, refiners, mask heads
  (unet.py, stacked_MRT.py, attentions.py, refinenet.py, feature_fusion.py, submodules.py:96-145)   -> K5 ``conv2d``
  (implicit GEMM on MFMA; torch.cat, bias, GELU/ReLU/sigmoid/tanh, residuals, GRU and FeatureFusion gates fused)
* GroupNorm / pre-norm LayerNorm                                                                     -> K6
* SelfAttn / CrossAttn incl. the contextual positional encoding (attentions.py:8-96)                  -> K4 ``attention``
* AvgPool2d / bilinear x2 (unet.py:25-37)                                                             -> ``resample2x``
* DispInit: LayerNorm + correlation, Sinkhorn OT, argmax + window regression (submodules.py:154-243) -> K1, K2
* CostVolume lookups (submodules.py:19-60)                                                            -> K3
* convex upsampling x4 / x1 (s2m2.py:101-133)                                                         -> K7
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import hip, pack

Tensor = torch.Tensor


class Spec(tuple):
    """(packed weight, packed bias, KH, KW, Cout padded) of one K5 layer; ``korder`` = K order of the packed weight (s2m2_conv2d)."""
    korder = 0


def pe_tables(h: int, w: int, device, pe_dim: int = 32) -> Tuple[Tensor, Tensor]:
    """Separable sinc relative positional-encoding tables of the reference's get_pe (utils.py:32-60): the dense
    pe[i, j, :] = 0.5 * [px[x_i - x_j + w - 1], py[y_i - y_j + h - 1]] is never materialised; K4 indexes the tables.
    Returns px (2w-1, 16), py (2h-1, 16), fp32."""
    def table(n: int) -> Tensor:
        L = 2 * n + 1
        sig = 5 / pe_dim
        pos = torch.linspace(-3, 3, L, device=device).tanh()
        dim_t = torch.linspace(-1, 1, pe_dim // 2, device=device)
        x = (dim_t[None, :] - pos[:, None]) / sig
        s = torch.where(x.abs() < 1e-6, torch.ones_like(x), torch.sin(3.1415 * x) / (3.1415 * x))
        return F.normalize(s, p=2, dim=-1)[:2 * n - 1].contiguous()
    return table(w), table(h)


MAX_PIXELS = 1 << 24            # K5 addresses N*H*W pixels of one tensor with 24 bits (conv.hip); the largest tensor is (2B,H,W,*)
LDS_BYTES = 160 * 1024


def check_limits(H: int, W: int, C: int, B: int = 1, dtype: Optional[torch.dtype] = None, use_pe: bool = True) -> None:
    """Geometry limits of the kernel library, validated before the first launch (they raise inside the C ABI otherwise).  With a
    dtype the attention of the feature pyramid at 1/32 (2B images, 8 heads of C/4 channels) is planned by the library itself: with
    ``use_pe`` (the checkpoint has ``enc3s.*.pe_proj``: positional-encoding variant, marginal-bin tiles + sinc tables next to the K/V
    stages in 160 KB of LDS, token grids up to 96 x 96), without it the plain key-split form, which has no grid limit.  Call it with the
    tensors' device current (the planner asks that device's LDS grant)."""
    if C not in (64, 128, 192, 256, 384):
        raise ValueError(f"feature_channels={C}: K1 is instantiated for 64, 128, 192, 256 and 384 channels")
    if 2 * H * W >= MAX_PIXELS:
        raise ValueError(f"{W}x{H}: one stereo pair must stay below 2^23 pixels per image (24-bit pixel index of the convolution kernels)")
    w = W // 4
    if (2 + 2 * 16) * (w + 1) * 4 > LDS_BYTES:
        raise ValueError(f"image width {W}: K2 keeps a cost-volume row's potentials in LDS, at most {4 * (LDS_BYTES // 136 - 1)} px wide")
    if dtype is not None:
        gh, gw = H // 32, W // 32
        ok, why = hip.attention_supported(2 * min(B, max_batch(H, W)), 8, gh * gw, C // 4, dtype, grid=(gw, gh) if use_pe else None)
        if not ok:
            what = "positional-encoding attention" if use_pe else "attention"
            raise ValueError(f"{W}x{H}: the {what} at 1/32 resolution does not take a {gw}x{gh} token grid ({why})")


def max_batch(H: int, W: int) -> int:
    """Pairs per launch sequence (S2M2.forward slices larger batches)."""
    return max(1, (MAX_PIXELS - 1) // (2 * H * W))


class Engine:
    def __init__(self, model, dtype: torch.dtype):
        hip.load()                                   # fail loudly if the kernel library is missing
        self.dtype = dtype
        self.C = model.feature_channels
        self.ntr = model.num_transformer
        self.use_positivity = model.use_positivity
        self.refine_iter = model.refine_iter
        self.output_upsample = model.output_upsample
        self.device = next(model.parameters()).device
        self.p: Dict[str, Tensor] = {n: q.detach().to(self.device, torch.float32) for n, q in model.named_parameters()}
        self._packed: Dict[object, Spec] = {}
        self._pe_cache: Dict[Tuple[int, int], Tuple[Tensor, Tensor]] = {}
        self._bufs: Dict[object, Tensor] = {}
        self._plans: Dict[object, object] = {}       # recorded refinement plans of EAGER calls (never evicted by zeros(): a plan owns the
        self._ns = None                              # tensors it returns); scratch namespace, see zeros()
        self._wsum: Dict[int, Tensor] = {}
        self._chain_ok: Dict[object, bool] = {}
        self._fusion_ok: Dict[object, bool] = {}
        self._wfrag = {}
        # One path per layer: which kernel form a layer takes follows from dtype and width alone (fp16 at C = 128 / 256: the direct forms of
        # K9 / K10 and the fragment-stream K5; other widths / fp32: the LDS-staged forms).  Two switches remain:
        # S2M2_FUSE_K1LN=0: K1 normalises the tokens itself (s2m2_ln_corr) instead of reading the LayerNorm output of the K9 launch that
        #   writes them -- bench.py reports both variants of the judged kernel (profiles/r04/ab_k1_fold.txt);
        # S2M2_CV_BAND=1 (opt-in, use_positivity models): banded cost volume, columns j <= i + 11.  Off by default: the reference's DispInit
        #   hands out the full unmasked volume (the captured "cv" stage), and K1 is 0.2 % of a forward
        self.fuse_k1ln = os.environ.get("S2M2_FUSE_K1LN", "1") != "0"
        self.cv_band = 11 if (os.environ.get("S2M2_CV_BAND", "0") == "1" and self.use_positivity) else -1
        # S2M2_K12=0: the narrow-input / few-cout 3x3 and 5x5 layers on K5's LDS-staged tiles instead of the pixel-split direct form
        # (A/B switch: profiles/r04/ab_conv_narrow.txt)
        self.use_k12 = os.environ.get("S2M2_K12", "1") != "0"
        # S2M2_K12_HEAD=0: UpsampleMask1x's conv_concat.0 and conv_concat.2 as two launches (K12 + K11) instead of the fused head
        self.use_k12_head = self.use_k12 and os.environ.get("S2M2_K12_HEAD", "1") != "0"
        # S2M2_REFINE_NATIVE=0: every refinement iteration enqueued from Python (A/B; the default replays a recorded plan: s2m2_refine_step)
        self.native_refine = os.environ.get("S2M2_REFINE_NATIVE", "1") != "0"
        # S2M2_ROWFUSE=0: every 1-D attention step as the launch triple Q|K|V fan-out / K4 / K9 chain (as up to round 5) instead of ONE K13
        # launch per step (hip.row_attn: fp16, C = 128 -- the S model's 1/4 and 1/8 levels; profiles/r06/ab_rowfuse.txt)
        self.use_rowfuse = os.environ.get("S2M2_ROWFUSE", "1") != "0"
        # S2M2_COARSE_FUSE=0 (A/B): the 1/32 level's down_conv and the first block's Q | K | V, and the last block's chain and the decoder's
        # up_conv, as separate launches (as up to round 5) instead of stages of one K9 launch each (profiles/r06/ab_coarse_fuse.txt)
        self.coarse_fuse = os.environ.get("S2M2_COARSE_FUSE", "1") != "0"
        # S2M2_CONVBLOCK=0 (A/B): every ConvBlock2D as three launches (K9 chain, K5 v5, K5 v5 + residual) instead of ONE K14 launch on the coarse
        # grids (at most S2M2_CONVBLOCK_MAXPIX pixels per launch; fp16, C = 128 / 256; profiles/r06/ab_convblock.txt)
        self.use_convblock = os.environ.get("S2M2_CONVBLOCK", "1") != "0"
        self.convblock_maxpix = int(os.environ.get("S2M2_CONVBLOCK_MAXPIX", "40000"))
        # (C = 256 -- the 1/16 level -- stays on the three launches: a block must produce all 256 intermediate channels, 96 blocks of 8 waves carry
        # the whole layer pair and the launch is MFMA-bound on a third of the chip: 50 vs 37 us, profiles/r06/convblock_bench.txt; S2M2_CONVBLOCK_C256=1)
        self.convblock_c256 = os.environ.get("S2M2_CONVBLOCK_C256", "0") == "1"
        # S2M2_GRU_FRAG=1 (A/B): ConvGRU's candidate layer (two-operand blend epilogue) on K5 v5's 64-pixel blocks instead of the v3 halo tiles
        self.gru_frag = os.environ.get("S2M2_GRU_FRAG", "0") == "1"
        self._tokens_normed: Optional[Tensor] = None             # DispInit's LayerNorm of feature_tr_4x, written by the last K9 launch
        self.ln_w = self.p["disp_init.layer_norm.weight"].contiguous()
        self.ln_b = self.p["disp_init.layer_norm.bias"].contiguous()
        self.k1_events = None                        # bench.py: list collecting (start, end) HIP events around K1

    # ---- weight packing (once per engine) ------------------------------------------------------------
    def std(self, name: str, splits: Optional[Sequence[Tuple[int, int]]] = None, transposed: bool = False, frag: bool = True) -> Spec:
        """One nn.Conv2d / nn.Linear (or stride-1 nn.ConvTranspose2d when ``transposed``) as a K5 weight.  frag=False: the layer is
        launched with a stride (or its epilogue needs two operands: the GRU blend stays on the v3 tiles), keep K order 0."""
        key = (name, tuple(splits) if splits else None, transposed, bool(frag))     # frag selects the packing (K order 0 / 2)
        s = self._packed.get(key)
        if s is None:
            w = self.p[name + ".weight"]
            if transposed:
                w = pack.convT_s1_as_conv(w)
            if w.dim() == 2:
                w = w[:, :, None, None]
            cin_p = sum(pd for _, pd in splits) if splits else pack.pad8(w.shape[1])
            frag = frag and pack.frag_eligible(pack.pad8(w.shape[0]), cin_p, w.shape[2], w.shape[3], self.dtype)
            wp = pack.pack_conv_frag(w, self.dtype, splits) if frag else pack.pack_conv(w, self.dtype, splits)
            s = Spec((wp, pack.pack_bias(self.p.get(name + ".bias"), w.shape[0]), w.shape[2], w.shape[3], pack.pad8(w.shape[0])))
            s.korder = 2 if frag else 0                            # spatial layers of wide tensors: weights as an MFMA fragment stream
            self._packed[key] = s
        return s

    def merged(self, key: str, parts: Sequence[Tuple[str, int, float, bool]], cin_total: int) -> Spec:
        """Several layers that read the same input tensor(s) stacked along Cout (each padded to 8 rows), every layer looking at its
        own channel window [offset, offset+Cin_i) of the combined input.  parts: (name, cin_offset, weight_scale, transposed)."""
        s = self._packed.get(key)
        if s is None:
            rows, biases = [], []
            kh = kw = 1
            for name, off, wscale, tr in parts:
                w = self.p[name + ".weight"]
                if tr:
                    w = pack.convT_s1_as_conv(w)
                if w.dim() == 2:
                    w = w[:, :, None, None]
                co, ci, kh, kw = w.shape
                cop = pack.pad8(co)
                full = w.new_zeros(cop, kh, kw, cin_total)
                full[:co, :, :, off:off + ci] = w.permute(0, 2, 3, 1) * wscale
                rows.append(full.reshape(cop, -1))
                bb = w.new_zeros(cop)
                if (name + ".bias") in self.p:
                    bb[:co] = self.p[name + ".bias"]
                biases.append(bb)
            wp = torch.cat(rows, 0)
            s = Spec((wp.to(self.dtype).contiguous(), torch.cat(biases).float().contiguous(), kh, kw, wp.shape[0]))
            if pack.frag_eligible(wp.shape[0], cin_total, kh, kw, self.dtype):
                w4 = wp.reshape(wp.shape[0], kh, kw, cin_total).permute(0, 3, 1, 2)            # back to (Cout, Cin, KH, KW): rows are padded already
                s = Spec((pack.pack_conv_frag(w4, self.dtype, [(cin_total, cin_total)]), s[1], kh, kw, wp.shape[0]))
                s.korder = 2
            self._packed[key] = s
        return s

    def convT2(self, name: str) -> Tuple[Spec, int]:
        """nn.ConvTranspose2d(kernel 2, stride 2) as the pixel-shuffle GEMM of K5."""
        key = (name, "T2")
        s = self._packed.get(key)
        if s is None:
            w = self.p[name + ".weight"]
            wp, cp = pack.pack_convT_2x2s2(w, self.dtype)
            s = Spec((wp, pack.pack_bias_shuffle(self.p.get(name + ".bias"), w.shape[1]), 1, 1, 4 * cp))
            self._packed[key] = s
        return s, s[4] // 4

    def cconv(self, spec: Spec, srcs, ln: bool = False, **kw) -> Tensor:
        """K5 launch.  ln: the layer is a pre-LayerNorm (no affine, attentions.py:117) followed by this 1x1 layer, folded into
        the kernel -- needs the row sums of the packed weight, computed once per layer."""
        wp, bp, kh, kw_, cout = spec
        # a plain 1x1 C -> C layer (C = 128 / 256, fp16, optionally + residual): the direct form of K9 as a one-stage chain -- the weight as
        # MFMA fragments straight into the operand registers instead of K5's LDS-staged K tiles
        if (kh == 1 and kw_ == 1 and len(srcs) == 1 and srcs[0].shape[-1] == cout and tuple(wp.shape) == (cout, cout)
                and not getattr(spec, "korder", 0) and set(kw) <= {"act", "epi", "aux0"}
                and kw.get("act", hip.ACT_NONE) in (hip.ACT_NONE, hip.ACT_GELU, hip.ACT_RELU)
                and kw.get("epi", hip.EPI_NONE) in (hip.EPI_NONE, hip.EPI_ADD) and ("aux0" in kw) == (kw.get("epi", hip.EPI_NONE) == hip.EPI_ADD)
                and ("aux0" not in kw or tuple(kw["aux0"].shape) == tuple(srcs[0].shape)) and self.chain_frag_ok(cout)):
            st = [(self.wfrag(spec), bp, kw.get("act", hip.ACT_NONE), self.wsum(spec) if ln else None)]
            if "aux0" in kw:
                return hip.mlp_chain(srcs[0], st, res=kw["aux0"], res_stage=0, frag=True)
            return hip.mlp_chain(srcs[0], st, frag=True)
        # any other plain 1x1 layer (rectangular, up to four concatenated sources, ConvTranspose 2x2 s2 included): K11, the direct form
        # (profiles/r04/pwbench.txt: 1.3 - 2.4 x the K5 launch per layer; ab_pw_direct.txt: 8.87 vs 9.00 ms per pair)
        if (kh == 1 and kw_ == 1 and not ln and not getattr(spec, "korder", 0) and len(srcs) <= 4 and set(kw) <= {"act", "shuffle2"}
                and kw.get("act", hip.ACT_NONE) in (hip.ACT_NONE, hip.ACT_GELU, hip.ACT_RELU)
                and wp.shape[1] == sum(t.shape[-1] for t in srcs) and self.pw_ok(wp.shape[1], cout)):
            return hip.pw_direct(srcs, self.wpw(spec), bp, cout, act=kw.get("act", hip.ACT_NONE), shuffle2=kw.get("shuffle2", 0))
        # spatial layers K5 would run on its LDS-staged tiles -- on an 8- / 16-channel tensor (the (disp, rgb) / (disp, conf, occ) side inputs,
        # the stem's output), or with few output channels (the mask / update heads, disp_feat.2): K12, the pixel-split direct form
        # (profiles/r04/narrowbench.txt)
        if (kh > 1 and not ln and not getattr(spec, "korder", 0) and len(srcs) <= 2 and srcs[0].dim() == 4 and set(kw) <= {"act", "stride"}
                and kw.get("act", hip.ACT_NONE) in (hip.ACT_NONE, hip.ACT_GELU, hip.ACT_RELU)
                and wp.shape[1] == kh * kw_ * sum(t.shape[-1] for t in srcs)
                and self.narrow_ok(kh, kw_, kw.get("stride", 1), sum(t.shape[-1] for t in srcs), cout)):
            return hip.conv_narrow(srcs, self.wnarrow(spec, kh * kw_), bp, kh, kw_, cout, stride=kw.get("stride", 1), act=kw.get("act", hip.ACT_NONE))
        if ln:
            kw["ln_wsum"] = self.wsum(spec)
        if getattr(spec, "korder", 0):
            kw["korder"] = spec.korder
        return hip.conv2d(srcs, wp, bp, kh, kw_, cout, **kw)

    def zeros(self, key, shape, dtype=None) -> Tensor:
        """Persistent zero-initialised scratch (padding channels stay zero; the live channels are rewritten by every use)."""
        # eager calls: one set per stream (re-entrant across streams); a GraphRunner swaps in its own dictionary (_ns set)
        ns = self._ns if (self._ns is not None or self.device.type != "cuda") else torch.cuda.current_stream(self.device).cuda_stream
        k = (key, tuple(shape), dtype or self.dtype, ns)
        b = self._bufs.get(k)
        if b is None:
            if self._ns is None and len(self._bufs) >= 64:        # eager scratch is per stream: callers that cycle streams must not leak a
                self._bufs.pop(next(iter(self._bufs)))            # set per stream -- oldest entry out (a GraphRunner owns its own dictionary)
            b = torch.zeros(shape, device=self.device, dtype=dtype or self.dtype)
            self._bufs[k] = b
        return b

    # ---- building blocks -----------------------------------------------------------------------------
    def down(self, p: str, x: Tensor) -> Tensor:
        """nn.AvgPool2d(2) -> Conv2d 1x1 (unet.py:24-29, stacked_MRT.py:21-26): the pooling is folded into the GEMM's operand load (same
        mean, rounded to the activation dtype like the stand-alone K7 launch it replaces)."""
        spec = self.std(p + ".1")
        c = x.shape[-1]
        pooled = spec[2] == 1 and spec[3] == 1 and x.dim() == 4 and x.shape[1] >= 2 and x.shape[2] >= 2
        if (pooled and tuple(spec[0].shape) in ((c, c), (2 * c, c)) and spec[4] == spec[0].shape[0] and not getattr(spec, "korder", 0)
                and self.chain_frag_ok(c)):
            # pooled 1x1 C -> C / C -> 2C: a fan-out-only launch of the direct K9 form, the 2x2 mean formed while the row tile is loaded
            return hip.mlp_fan(x, self.wfrag(spec), spec[1], None, pool2=True)
        if pooled:
            return self.cconv(spec, [x], pool2=True)
        return self.cconv(spec, [hip.resample2x(x, 0)])

    def up(self, p: str, x: Tensor) -> Tensor:
        """nn.Upsample(bilinear x2) -> Conv2d 1x1 (unet.py:32-37, stacked_MRT.py:29-34).  A 1x1 layer commutes with the bilinear
        resampling (per-pixel interpolation weights sum to 1, so the bias passes through too): the GEMM runs on the coarse grid
        -- a quarter of the pixels -- and K7 resamples Cout channels instead of Cin."""
        spec = self.std(p + ".1")
        if spec[2] == 1 and spec[3] == 1:
            return hip.resample2x(self.cconv(spec, [x]), 1)
        return self.cconv(spec, [hip.resample2x(x, 1)])

    def conv_block(self, p: str, z: Tensor) -> Tensor:
        """ConvBlock2D (attentions.py:255-281): conv3-GELU-conv3 + conv1-ReLU-conv1."""
        c0, c2 = self.std(p + ".convs_1x.0"), self.std(p + ".convs_1x.2")
        c = z.shape[-1]
        same = c0[4] == c and c2[4] == c
        if (self.use_convblock and same and z.dim() == 4 and z.shape[0] * z.shape[1] * z.shape[2] <= self.convblock_maxpix
                and (c == 128 or self.convblock_c256) and self.chain_frag_ok(c) and self.convblock_ok(c, z.shape[1], z.shape[2])):
            k0, k2 = self.std(p + ".convs.0"), self.std(p + ".convs.2")
            if (getattr(k0, "korder", 0) == 2 and getattr(k2, "korder", 0) == 2 and k0[2] == 3 and k0[3] == 3 and k2[2] == 3 and k2[3] == 3
                    and k0[4] == c and k2[4] == c and k0[0].numel() == 9 * c * c and k2[0].numel() == 9 * c * c
                    and tuple(c0[0].shape) == (c, c) and tuple(c2[0].shape) == (c, c)):
                # the whole block -- 1x1 branch, 3x3 - GELU - 3x3, the final add -- as ONE K14 launch (the coarse grids: latency chains)
                return hip.conv_block(z, k0[0], k0[1], k2[0], k2[1], self.wfrag(c0), c0[1], self.wfrag(c2), c2[1])
        if same and self.chain_frag_ok(c):                         # the 1x1 branch as one K9 launch (direct form)
            b = hip.mlp_chain(z, [(self.wfrag(c0), c0[1], hip.ACT_RELU, None), (self.wfrag(c2), c2[1], hip.ACT_NONE, None)], frag=True)
        elif same and self.chain_ok(c):
            b = hip.mlp_chain(z, [(c0[0], c0[1], hip.ACT_RELU, None), (c2[0], c2[1], hip.ACT_NONE, None)])
        else:
            b = self.cconv(c2, [self.cconv(c0, [z], act=hip.ACT_RELU)])
        t = self.cconv(self.std(p + ".convs.0"), [z], act=hip.ACT_GELU)
        return self.cconv(self.std(p + ".convs.2"), [t], epi=hip.EPI_ADD, aux0=b)

    def dual_heads(self, p: str):
        """[gate.2 | fusion.2] stacked along K (the channel order of the hidden tensor) + the two biases"""
        key = p + "|gate.2+fusion.2"
        dual = self._packed.get(key)
        if dual is None:
            g2, f2 = self.std(p + ".feature_gate.2"), self.std(p + ".feature_fusion.2")
            dual = self._packed[key] = (torch.cat([g2[0], f2[0]], dim=1).contiguous(), g2[1], f2[1])
        return dual

    def fusion_ok(self, c: int) -> bool:
        """K10 exists for this width in one of its forms (LDS-staged: 128 / 256 in both dtypes; direct: fp16 128 / 192 / 256 / 384)"""
        ok = self._fusion_ok.get(c)
        if ok is None:
            ok = self._fusion_ok[c] = hip.feature_fusion_supported(c, self.dtype) or self.fusion_frag_ok(c)
        return ok

    def k10(self, p: str, z0: Tensor, z1: Tensor, first: Spec, z1_coarse: bool = False) -> Tensor:
        """one K10 launch; fp16 at C = 128 / 256 takes the direct form (weights as one fragment stream, permuted once per layer)"""
        dual = self.dual_heads(p)
        c = z0.shape[-1]
        if self.fusion_frag_ok(c):
            key = p + "|k10 fragment stream"
            ws = self._packed.get(key)
            if ws is None:
                ws = self._packed[key] = pack.fusion_frag(first[0], dual[0])
            return hip.feature_fusion(z0, z1, ws, first[1], None, dual[1], dual[2], z1_coarse=z1_coarse, frag=True)
        return hip.feature_fusion(z0, z1, first[0], first[1], dual[0], dual[1], dual[2], z1_coarse=z1_coarse)

    def fusion_up_ok(self, p: str, c: int, pu: str) -> bool:
        """the decoder's fusion takes the coarse-grid form (1x1 up_conv on the coarse grid, K10 reads it through the bilinear resampling)"""
        spec = self.std(pu + ".1")
        first = self.merged(p + "|gate+fusion", [(p + ".feature_gate.0", 0, 1.0, False), (p + ".feature_fusion.0", 0, 1.0, False)], 2 * c)
        return (spec[2] == 1 and spec[3] == 1 and spec[4] == c and first[2] == 1 and first[3] == 1
                and self.p[p + ".feature_gate.0.weight"].shape[0] == c and self.fusion_ok(c))

    def up_tail(self, p: str, c_fine: int, pu: str, c_coarse: int) -> Optional[Spec]:
        """the decoder's up_conv as a fan-out stage of the K9 launch that produces its input (attn_ffn ``tail``): a plain C -> C 1x1 layer in
        front of a coarse-grid fusion, direct K9 form available"""
        spec = self.std(pu + ".1")
        if (self.coarse_fuse and self.fusion_up_ok(p, c_fine, pu) and tuple(spec[0].shape) == (c_coarse, c_coarse) and c_fine == c_coarse
                and not getattr(spec, "korder", 0) and self.chain_frag_ok(c_coarse)):
            return spec
        return None

    def fusion_up(self, p: str, z0: Tensor, pu: str, xc: Tensor, up_pre: Optional[Tensor] = None) -> Tensor:
        """fusion(z0, up_conv(xc)) of the decoders (unet.py:98-110, stacked_MRT.py:113-121): the 1x1 up_conv runs on the coarse grid
        (see up()) and K10 reads its output through the bilinear resampling -- no stand-alone K7 launch, no upsampled tensor.
        up_pre: up_conv(xc) on the coarse grid, where the launch that produced xc computed it (up_tail)."""
        spec = self.std(pu + ".1")
        c = z0.shape[-1]
        first = self.merged(p + "|gate+fusion", [(p + ".feature_gate.0", 0, 1.0, False), (p + ".feature_fusion.0", 0, 1.0, False)], 2 * c)
        if self.fusion_up_ok(p, c, pu):
            return self.k10(p, z0, up_pre if up_pre is not None else self.cconv(spec, [xc]), first, z1_coarse=True)
        return self.fusion(p, z0, self.up(pu, xc))

    def fusion(self, p: str, z0: Tensor, z1: Tensor) -> Tensor:
        """FeatureFusion (feature_fusion.py:4-33): out = fusion(z) + g*z0 + (1-g)*z1, g = clamp(sigmoid(gate(z)), .01, .99).
        Both first layers read cat(z0, z1): one GEMM; the gate mix and the final add are epilogues."""
        c = z0.shape[-1]
        cg = self.p[p + ".feature_gate.0.weight"].shape[0]
        spec = self.merged(p + "|gate+fusion", [(p + ".feature_gate.0", 0, 1.0, False), (p + ".feature_fusion.0", 0, 1.0, False)], 2 * c)
        if spec[2] == 1 and spec[3] == 1 and cg == c and self.fusion_ok(c):
            return self.k10(p, z0, z1, spec)                       # K10: the whole block in one launch, h never leaves the CU
        gf = self.cconv(spec, [z0, z1], act=hip.ACT_GELU)
        if cg % 64 == 0:
            # both second layers in one launch: weight rows [gate.2 | fusion.2] along K (= the channel order of gf), two accumulators
            dual = self.dual_heads(p)
            return hip.conv2d([gf], dual[0], dual[1], 1, 1, dual[0].shape[0], act=hip.ACT_SIGMOID, epi=hip.EPI_DUALMIX, aux0=z0, aux1=z1,
                              ksplit=cg, bias2=dual[2])
        m = self.cconv(self.std(p + ".feature_gate.2"), [gf[..., :cg]], act=hip.ACT_SIGMOID, epi=hip.EPI_GATEMIX, aux0=z0, aux1=z1)
        return self.cconv(self.std(p + ".feature_fusion.2"), [gf[..., cg:]], epi=hip.EPI_ADD, aux0=m)

    # ---- attention -----------------------------------------------------------------------------------
    def pe(self, h: int, w: int) -> Tuple[Tensor, Tensor]:
        key = (h, w)
        if key not in self._pe_cache:
            self._pe_cache[key] = pe_tables(h, w, self.device)
        return self._pe_cache[key]

    def qkv_spec(self, p: str) -> Spec:
        """[q | k | v] of one attention module (attentions.py:24-28,71-74) stacked along Cout: one GEMM"""
        c = self.p[p + ".q.weight"].shape[1]
        return self.merged(p + "|qkv", [(p + ".q", 0, 1.0, False), (p + ".k", 0, 1.0, False), (p + ".v", 0, 1.0, False)], c)

    def qkv(self, p: str, x: Tensor) -> Tensor:
        spec = self.qkv_spec(p)
        c = x.shape[-1]
        if spec[2] == 1 and spec[3] == 1 and spec[4] == 3 * c and self.chain_frag_ok(c):
            return hip.mlp_fan(x, self.wfrag(spec), spec[1], self.wsum(spec))     # direct form: fragments straight into registers
        return self.cconv(spec, [x], ln=True)                                    # pre-LayerNorm folded into the K5 launch

    def attn_core(self, p: str, z: Tensor, nh: int, two_d: bool, cross: bool, use_pe: bool, qkv: Optional[Tensor] = None) -> Tensor:
        """pre-LN -> fused QKV projection -> K4; returns the attention output BEFORE the output projection (see attn_ffn).  qkv: the
        projection of ``z`` when the K9 launch that produced ``z`` already computed it (attn_ffn's fan-out stages)."""
        n, h, w, c = z.shape
        if qkv is None:
            qkv = self.qkv(p + ".attn", z)                   # pre-LN folded into the projection
        v3 = qkv.reshape(n, h * w, 3 * c) if two_d else qkv.reshape(n * h, w, 3 * c)
        q, k, v = v3[..., :c], v3[..., c:2 * c], v3[..., 2 * c:]
        if use_pe:
            px, py = self.pe(h, w)
            o, pes = hip.attention(q, k, v, nh, pe=(px, py, w, h))
            d = c // nh
            o = self.cconv(self.std(p + ".attn.pe_proj"), [pes.reshape(1, 1, -1, 32)], epi=hip.EPI_ADD, aux0=o.reshape(1, 1, -1, d))
        else:
            o = hip.attention(q, k, v, nh, swap_halves=cross)
        return o.reshape(n, h, w, c)

    def wsum(self, spec: Spec) -> Tensor:
        """row sums of a packed weight (the pre-LayerNorm correction term of K5 / K9), computed once per layer"""
        wp = spec[0]
        ws = self._wsum.get(wp.data_ptr())
        if ws is None:
            ws = wp.float().sum(dim=1).contiguous()
            self._wsum[wp.data_ptr()] = ws
        return ws

    def wfrag(self, spec: Spec) -> Tensor:
        """a packed 1x1 weight in the MFMA-fragment order of the direct K9 form (pack.chain_frag), permuted once per layer"""
        wp = spec[0]
        wf = self._wfrag.get(wp.data_ptr())
        if wf is None:
            wf = self._wfrag[wp.data_ptr()] = pack.chain_frag(wp)
        return wf

    def attn_ffn(self, pa: str, pf: str, o: Tensor, z: Tensor, ln_out=None, next_attn: Optional[str] = None, tail: Optional[Spec] = None):
        """z' = z + proj(o);  z' + ffn.2(GELU(ffn.0(LayerNorm(z')))) (attentions.py:311-321,347-355): one K9 launch when the width
        is supported, else three K5 launches (pre-LN folded into the first FFN layer).  ln_out = (gamma, beta, eps): the K9 launch also
        writes LayerNorm(result) * gamma + beta (kept in ``self._tokens_normed`` for K1, see features()).  next_attn: prefix of the
        attention module that reads the result next -- its pre-LN + Q | K | V projection runs as fan-out stages of the same K9 launch
        (the rows are still in LDS): one launch and one round trip of the rows less per attention.  -> (result, qkv or None)."""
        c = z.shape[-1]
        proj, f0, f2 = self.std(pa + ".attn.proj"), self.std(pf + ".ffn.0"), self.std(pf + ".ffn.2")
        if ln_out is not None and not hip.mlp_chain_ln_out_supported(c, self.dtype):
            ln_out = None                                          # (widths without the LayerNorm output: K1 normalises the tokens itself)
        # K1 places image row y on XCD y / (h / 8): hand the token rows of that eighth of every image to the same XCD
        n, h, w, _ = z.shape
        grp = ((h // 8) * w if h % 8 == 0 else 0) if ln_out is not None else 0
        acts = (hip.ACT_NONE, hip.ACT_GELU, hip.ACT_NONE)
        if self.chain_frag_ok(c):
            # direct form: fragments straight into registers; the next attention's Q | K | V projection rides along as fan-out stages (one
            # launch and one round trip of the rows less), or the launch that writes feature_tr_4x also writes K1's normalised tokens
            st = [(self.wfrag(sp), sp[1], act, self.wsum(sp) if sp is f0 else None) for sp, act in zip((proj, f0, f2), acts)]
            if ln_out is not None:
                out, self._tokens_normed = hip.mlp_chain(o, st, res=z, res_stage=0, carry=True, ln_out=ln_out, xcd_group_rows=grp, frag=True)
                return out, None
            if next_attn is not None:
                qs = self.qkv_spec(next_attn)
                if qs[2] == 1 and qs[3] == 1 and qs[4] == 3 * c:
                    return hip.mlp_chain(o, st, res=z, res_stage=0, carry=True, fan=(self.wfrag(qs), qs[1], self.wsum(qs)), frag=True)
            if tail is not None:
                # the layer applied to the result next (the decoder's 1x1 up_conv on the coarse grid) as a fan-out stage of this launch: no
                # LayerNorm fold, its own bias -> (result, up_conv(result))
                return hip.mlp_chain(o, st, res=z, res_stage=0, carry=True, fan=(self.wfrag(tail), tail[1], None), frag=True)
            return hip.mlp_chain(o, st, res=z, res_stage=0, carry=True, frag=True), None
        if self.chain_ok(c):                                       # LDS-staged form (fp32; fp16 at C = 384 / 512)
            st = [(sp[0], sp[1], act, self.wsum(sp) if sp is f0 else None) for sp, act in zip((proj, f0, f2), acts)]
            if ln_out is not None:
                out, self._tokens_normed = hip.mlp_chain(o, st, res=z, res_stage=0, carry=True, ln_out=ln_out, xcd_group_rows=grp)
                return out, None
            return hip.mlp_chain(o, st, res=z, res_stage=0, carry=True), None
        z = self.cconv(proj, [o], epi=hip.EPI_ADD, aux0=z)         # widths K9 does not take (C = 192): three K5 launches
        hdn = self.cconv(f0, [z], ln=True, act=hip.ACT_GELU)
        return self.cconv(f2, [hdn], epi=hip.EPI_ADD, aux0=z), None

    def chain_frag_ok(self, c: int, dtype=None) -> bool:
        """the direct form of K9 exists for this width (asked once per width: the query is a ctypes call)"""
        ok = self._chain_ok.get(("frag", c))
        if ok is None:
            ok = self._chain_ok[("frag", c)] = hip.mlp_chain_frag_supported(c, self.dtype)
        return ok

    def fusion_frag_ok(self, c: int) -> bool:
        ok = self._fusion_ok.get(("frag", c))
        if ok is None:
            ok = self._fusion_ok[("frag", c)] = hip.feature_fusion_frag_supported(c, self.dtype)
        return ok

    def pw_ok(self, k: int, cout: int) -> bool:
        """K11 (hip.pw_direct) exists for a (cout, k) 1x1 layer in this dtype (asked once per shape)"""
        ok = self._chain_ok.get(("pw", k, cout))
        if ok is None:
            ok = self._chain_ok[("pw", k, cout)] = hip.pw_direct_supported(k, cout, self.dtype)
        return ok

    def narrow_ok(self, kh: int, kw: int, stride: int, cin: int, cout: int) -> bool:
        """K12 (hip.conv_narrow) exists for this spatial layer on a cin-channel tensor in this dtype (asked once per shape)"""
        key = ("narrow", kh, kw, stride, cin, cout)
        ok = self._chain_ok.get(key)
        if ok is None:
            ok = self._chain_ok[key] = self.use_k12 and hip.conv_narrow_supported(kh, kw, stride, cin, cout, self.dtype)
        return ok

    def wnarrow(self, spec: Spec, ntap: int) -> Tensor:
        """a packed spatial weight in the fragment order of K12 (pack.narrow_frag), permuted once per layer"""
        wp = spec[0]
        wf = self._wfrag.get(("narrow", wp.data_ptr()))
        if wf is None:
            wf = self._wfrag[("narrow", wp.data_ptr())] = pack.narrow_frag(wp, ntap)
        return wf

    def wpw(self, spec: Spec) -> Tensor:
        """a packed 1x1 weight in the fragment order of K11 (pack.pw_frag), permuted once per layer"""
        wp = spec[0]
        wf = self._wfrag.get(("pw", wp.data_ptr()))
        if wf is None:
            wf = self._wfrag[("pw", wp.data_ptr())] = pack.pw_frag(wp)
        return wf

    def chain_ok(self, c: int) -> bool:
        ok = self._chain_ok.get(c)
        if ok is None:
            ok = self._chain_ok[c] = hip.mlp_chain_supported(c, self.dtype)
        return ok

    def first_attn(self, p: str) -> str:
        """prefix of the attention module a block applies first (cross attention where the block has one)"""
        return p + (".cross_attn.attn" if (p + ".cross_attn.attn.q.weight") in self.p else ".self_attn.attn")

    def convblock_ok(self, c: int, h: int, w: int) -> bool:
        """K14 takes a ConvBlock2D of this width on this grid (asked once per shape)"""
        key = ("convblock", c, h, w)
        ok = self._chain_ok.get(key)
        if ok is None:
            ok = self._chain_ok[key] = hip.conv_block_supported(c, h, w, self.dtype)
        return ok

    def row_ok(self, z: Tensor, nh: int) -> bool:
        """K13 takes the 1-D attention steps on this tensor (asked once per shape)"""
        key = ("row", z.shape[-1], nh, z.shape[2])
        ok = self._chain_ok.get(key)
        if ok is None:
            ok = self._chain_ok[key] = self.use_rowfuse and z.dim() == 4 and hip.row_attn_supported(z.shape[-1], nh, z.shape[2], self.dtype)
        return ok

    def row_step(self, pa: str, pf: str, z: Tensor, nh: int, cross: bool, ln_out=None) -> Tensor:
        """One 1-D attention step -- pre-LN, Q | K | V, attention along the row (against the other view's row when ``cross``), proj + residual,
        pre-LN, FFN + residual (attentions.py:131-161 / :99-128 with :229-250) -- as ONE K13 launch.  The step's six layers in the row_attn
        packing and its twelve per-channel vectors are packed once per step (pack.rowattn_pack / rowattn_vectors)."""
        key = ("rowstep", pa, pf, ln_out is not None)
        ent = self._packed.get(key)
        if ent is None:
            c = z.shape[-1]
            qs, proj, f0, f2 = self.qkv_spec(pa + ".attn"), self.std(pa + ".attn.proj"), self.std(pf + ".ffn.0"), self.std(pf + ".ffn.2")
            ws = self.wsum(qs)
            cut = lambda t: (None, None, None) if t is None else (t[:c], t[c:2 * c], t[2 * c:])      # noqa: E731
            wts = pack.rowattn_pack(torch.cat([qs[0], proj[0], f0[0], f2[0]], 0))
            vec = pack.rowattn_vectors(cut(ws) + (self.wsum(f0),), cut(qs[1]) + (proj[1], f0[1], f2[1]), ln_out[:2] if ln_out is not None else None)
            ent = self._packed[key] = (wts, vec)
        if ln_out is not None:
            out, self._tokens_normed = hip.row_attn(z, nh, cross, ent[0], ent[1], ln_out_eps=ln_out[2])
            return out
        return hip.row_attn(z, nh, cross, ent[0], ent[1])

    def attn_block(self, p: str, z: Tensor, nh: int, two_d: bool, use_pe: bool = False, ln_out=None, qkv_in: Optional[Tensor] = None,
                   next_block: Optional[str] = None, tail: Optional[Spec] = None):
        """BasicAttnBlock (1-D, attentions.py:347-355) / GlobalAttnBlock (2-D, :311-321).  ln_out: see attn_ffn (last launch of the block).
        qkv_in: the Q | K | V projection of ``z`` for the block's first attention, if the launch that produced ``z`` computed it;
        next_block: prefix of the attention block applied to the result next (its first projection is computed here).
        -> (result, that projection or None)."""
        if not two_d and not use_pe and qkv_in is None and next_block is None and self.row_ok(z, nh):
            if (p + ".cross_attn.attn.q.weight") in self.p:
                z = self.row_step(p + ".cross_attn", p + ".ffn_c", z, nh, True)
            return self.row_step(p + ".self_attn", p + ".ffn", z, nh, False, ln_out=ln_out), None
        if (p + ".cross_attn.attn.q.weight") in self.p:
            o = self.attn_core(p + ".cross_attn", z, nh, two_d, True, False, qkv=qkv_in)
            z, qkv_in = self.attn_ffn(p + ".cross_attn", p + ".ffn_c", o, z, next_attn=p + ".self_attn.attn")
        o = self.attn_core(p + ".self_attn", z, nh, two_d, False, use_pe, qkv=qkv_in)
        return self.attn_ffn(p + ".self_attn", p + ".ffn", o, z, ln_out=ln_out, next_attn=self.first_attn(next_block) if next_block else None,
                             tail=tail if next_block is None else None)

    def _count(self, prefix: str) -> int:
        n = 0
        while f"{prefix}.{n}.ffn.ffn.0.weight" in self.p:
            n += 1
        return n

    # ---- U-Net / MRT ---------------------------------------------------------------------------------
    def unet(self, p: str, z: Tensor):
        use_pe = (p + ".enc3s.0.self_attn.attn.pe_proj.weight") in self.p
        z0 = self.conv_block(p + ".enc0", z)
        z1 = self.conv_block(p + ".enc1", self.down(p + ".down_conv0", z0))
        z2 = self.conv_block(p + ".enc2", self.down(p + ".down_conv1", z1))
        blocks = [(f"{p}.enc3s.{i}", use_pe) for i in range(self._count(p + ".enc3s"))] + \
                 [(f"{p}.dec3s.{i}", False) for i in range(self._count(p + ".dec3s"))]
        q = None
        dspec = self.std(p + ".down_conv2.1")
        c2 = z2.shape[-1]
        qs = self.qkv_spec(self.first_attn(blocks[0][0])) if blocks else None
        if (self.coarse_fuse and qs is not None and dspec[2] == 1 and dspec[3] == 1 and tuple(dspec[0].shape) == (c2, c2) and dspec[4] == c2
                and not getattr(dspec, "korder", 0) and z2.shape[1] >= 2 and z2.shape[2] >= 2 and self.chain_frag_ok(c2)
                and qs[2] == 1 and qs[3] == 1 and qs[4] == 3 * c2):
            # AvgPool2d(2) + down_conv2 (unet.py:24-29) and the first attention block's pre-LN + Q | K | V in ONE K9 launch: a one-stage chain
            # on the pooled tile with the projection as its fan-out stages
            z3, q = hip.mlp_chain(z2, [(self.wfrag(dspec), dspec[1], hip.ACT_NONE, None)], fan=(self.wfrag(qs), qs[1], self.wsum(qs)),
                                  frag=True, pool2=True)
        else:
            z3 = self.down(p + ".down_conv2", z2)
        tail = self.up_tail(p + ".concat_conv2", c2, p + ".up_conv2", z3.shape[-1]) if blocks else None
        up3 = None
        for k, (bp, pe) in enumerate(blocks):                     # consecutive blocks on the same tensor: each computes the next one's Q | K | V
            last = k + 1 == len(blocks)
            z3, q = self.attn_block(bp, z3, 8, True, pe, qkv_in=q, next_block=None if last else blocks[k + 1][0], tail=tail if last else None)
            if last and tail is not None:
                up3, q = q, None
        n2 = self.conv_block(p + ".dec2", self.fusion_up(p + ".concat_conv2", z2, p + ".up_conv2", z3, up_pre=up3))
        n1 = self.conv_block(p + ".dec1", self.fusion_up(p + ".concat_conv1", z1, p + ".up_conv1", n2))
        n0 = self.conv_block(p + ".dec0", self.fusion_up(p + ".concat_conv0", z0, p + ".up_conv0", n1))
        return n0, n1, n2, z3

    def mrt(self, p: str, z0: Tensor, z1: Tensor, z2: Tensor, z3: Tensor, ln_out=None):
        z0 = self.attn_block(p + ".enc_attn0", z0, 1, False)[0]
        z1 = self.attn_block(p + ".enc_attn1", self.fusion(p + ".down_concat1", z1, self.down(p + ".down_conv0", z0)), 2, False)[0]
        z2 = self.attn_block(p + ".enc_attn2", self.fusion(p + ".down_concat2", z2, self.down(p + ".down_conv1", z1)), 4, False)[0]
        z3 = self.fusion(p + ".down_concat3", z3, self.down(p + ".down_conv2", z2))
        blocks = [f"{p}.enc_attn3s.{i}" for i in range(2)] + [f"{p}.dec_attn3s.{i}" for i in range(2)]
        q = None
        tail = self.up_tail(p + ".up_concat2", z2.shape[-1], p + ".up_conv2", z3.shape[-1])
        up3 = None
        for k, bp in enumerate(blocks):
            last = k + 1 == len(blocks)
            z3, q = self.attn_block(bp, z3, 8, True, qkv_in=q, next_block=None if last else blocks[k + 1], tail=tail if last else None)
            if last and tail is not None:
                up3, q = q, None
        z2 = self.attn_block(p + ".dec_attn2", self.fusion_up(p + ".up_concat2", z2, p + ".up_conv2", z3, up_pre=up3), 4, False)[0]
        z1 = self.attn_block(p + ".dec_attn1", self.fusion_up(p + ".up_concat1", z1, p + ".up_conv1", z2), 2, False)[0]
        z0 = self.attn_block(p + ".dec_attn0", self.fusion_up(p + ".up_concat0", z0, p + ".up_conv0", z1), 1, False, ln_out=ln_out)[0]
        return z0, z1, z2, z3

    # ---- refiners ------------------------------------------------------------------------------------
    def global_refiner(self, p: str, ctx: Tensor, disp: Tensor, conf: Tensor) -> Tensor:
        """GlobalRefiner (refinenet.py:39-73) + the clamp of s2m2.py:160-161; disp, conf (B,1,h,w) fp32."""
        small = hip.refine_prep(disp, conf, None, 0, self.dtype)
        f = self.cconv(self.std(p + ".init_feat.0", splits=[(2, 8), (self.C, self.C)]), [small, ctx], act=hip.ACT_GELU)
        f = self.cconv(self.std(p + ".init_feat.2"), [f])
        f = self.unet(p + ".refine_unet", f)[0]
        return hip.global_update(self.cconv(self.std(p + ".out_feat.0"), [f]), disp, conf, self.use_positivity)

    def gru(self, p: str, h: Tensor, x: Tensor) -> Tensor:
        """ConvGRU (refinenet.py:7-36): two separable passes; the gate arithmetic lives in the conv epilogues.  z and r read the same
        cat(h, x) through layers of the same shape: ONE launch with the two layers stacked along Cout, whose upper half carries the
        r * h epilogue (s2m2_conv_desc.epi_cout0; fp16 / K order 2) -- 6 launches fewer per refinement step and cat(h, x) read once."""
        C = h.shape[-1]
        for sfx in ("1", "2"):
            zr = self.merged(f"{p}|zr{sfx}", [(f"{p}.convz{sfx}", 0, 1.0, False), (f"{p}.convr{sfx}", 0, 1.0, False)], h.shape[-1] + x.shape[-1])
            if getattr(zr, "korder", 0) == 2 and zr[4] == 2 * C and C % 128 == 0:
                both = self.cconv(zr, [h, x], act=hip.ACT_SIGMOID, epi=hip.EPI_MUL, aux0=h, epi_cout0=C)
                z, rh = both[..., :C], both[..., C:]
            else:
                z = self.cconv(self.std(f"{p}.convz{sfx}"), [h, x], act=hip.ACT_SIGMOID)
                rh = self.cconv(self.std(f"{p}.convr{sfx}"), [h, x], act=hip.ACT_SIGMOID, epi=hip.EPI_MUL, aux0=h)
            h = self.cconv(self.std(f"{p}.convq{sfx}", frag=self.gru_frag), [rh, x], act=hip.ACT_TANH, epi=hip.EPI_GRU, aux0=z, aux1=h)
        return h

    def local_refiner(self, p: str, hidden: Tensor, ctx: Tensor, disp: Tensor, conf: Tensor, occ: Tensor, cv: Tensor, cap, it,
                      small: Optional[Tensor] = None, want_small: bool = False):
        """LocalRefiner.forward (refinenet.py:126-154) + the loop epilogue of s2m2.py:177-180 (clamp, occlusion mask).  small: the
        (disp, conf, occ) side input if the previous iteration's epilogue already produced it; want_small: produce the next one."""
        B, _, h, w = disp.shape
        C = self.C
        # corr/16 -> 1x1(9->96) GELU 1x1(96->64), both levels as block-diagonal GEMMs (the 1/16 is folded into the weights)
        sa = self.merged(p + "|corrA", [(p + ".corr_feat1.0", 0, 1 / 16, False), (p + ".corr_feat2.0", 16, 1 / 16, False)], 32)
        sb = self.merged(p + "|corrB", [(p + ".corr_feat1.2", 0, 1.0, False), (p + ".corr_feat2.2", 96, 1.0, False)], 192)
        corr = self.zeros("lr_corr", (B, h, w, 32))
        hip.cv_lookup_into(cv, disp.contiguous(), corr, 0, 16, 4)                                        # K3
        f12 = self.cconv(sb, [self.cconv(sa, [corr], act=hip.ACT_GELU)])                                 # K11 x 2
        if cap is not None:
            # clones: lr_corr is persistent scratch that the next iteration overwrites
            cap[f"corr1_it{it}"], cap[f"corr2_it{it}"] = (corr[..., 0:9].permute(0, 3, 1, 2).clone(),
                                                          corr[..., 16:25].permute(0, 3, 1, 2).clone())
        if small is None:
            small = hip.refine_prep(disp, conf, occ, 1, self.dtype)
        dc = self.cconv(self.merged(p + "|dcA", [(p + ".disp_feat.0", 0, 1.0, False), (p + ".conf_occ_feat.0", 1, 1.0, False)], 8),
                        [small], act=hip.ACT_GELU)
        fd = self.cconv(self.std(p + ".disp_feat.2"), [dc[..., :96]])
        fc = self.cconv(self.std(p + ".conf_occ_feat.2"), [dc[..., 96:160]])
        x = self.cconv(self.std(p + ".disp_corr_ctx_cat.0"), [fd, f12, ctx, fc], act=hip.ACT_GELU)
        x = self.cconv(self.std(p + ".disp_corr_ctx_cat.2"), [x])
        x = self.unet(p + ".refine_unet", x)[0]
        hn = self.gru(p + ".gru", hidden, x)
        u = self.cconv(self.merged(p + "|updA", [(p + ".disp_update.0", 0, 1.0, False), (p + ".conf_occ_update.0", 0, 1.0, False)], C),
                       [hn], act=hip.ACT_GELU)
        dco = self.cconv(self.merged(p + "|updB", [(p + ".disp_update.2", 0, 1.0, False), (p + ".conf_occ_update.2", C, 1.0, False)], 2 * C),
                         [u])
        return (hn,) + tuple(hip.refine_update(dco, disp, conf, occ, self.use_positivity, want_small=want_small))

    def refine_step_native(self, it: int, hidden: Tensor, ctx: Tensor, disp: Tensor, conf: Tensor, occ: Tensor, cv: Tensor,
                           small: Optional[Tensor], want_small: bool):
        """One refinement iteration through the library's recorded plan (s2m2_refine_step, include/s2m2_hip.h): the ~55 launches of
        local_refiner are enqueued by ONE native call instead of ~55 Python -> ctypes round trips.  Life cycle per (iteration, shapes,
        scratch namespace): call 1 runs local_refiner as always (weights get packed, kernel attributes set), call 2 runs it once more while the
        library records (its intermediates and outputs are allocated from a private MemPool that lives with the plan), every later call
        replays the plan with the seven externals -- hidden, ctx, disp, conf, occ, cv, side input -- wherever they are now; the iteration's
        outputs are the tensors of the recorded run.  A captured hipGraph (GraphRunner: two warm-up forwards, then the capture) therefore
        holds the plan's launches; bit-identical to the Python-enqueued iteration (tests/test_hip_e2e.py)."""
        ns = self._ns if self._ns is not None else torch.cuda.current_stream(self.device).cuda_stream   # eager: one plan (and its intermediates) per stream
        key = ("refine_plan", it, tuple(hidden.shape), tuple(cv.shape), cv.stride(2), hidden.dtype, small is None, want_small, ns)
        # a GraphRunner's plans live (and die) with its scratch dictionary; eager plans in their own dictionary, which zeros() never evicts
        # (an evicted plan's result tensors could still be the live hidden / disp of the running forward).  NOTE (aliasing, as with a
        # hipGraph): the iteration's outputs are the tensors of the recorded run -- the next forward on the same stream overwrites them
        # in place; S2M2.forward hands out clones / freshly upsampled maps, callers of Engine.finish must not keep these across forwards
        store = self._bufs if self._ns is not None else self._plans
        ent = store.get(key)
        if ent is None:                                            # first call: the plain Python path (one-time set-up happens here)
            if store is self._plans and len(store) >= 32:
                store.pop(next(iter(store)))                       # callers cycling through streams: oldest plan out (its tensors stay
            store[key] = "warm"                                    # alive as long as somebody holds them)
            return self.local_refiner("refiner", hidden, ctx, disp, conf, occ, cv, None, it, small=small, want_small=want_small)
        ext = [hidden, ctx, disp, conf, occ, cv, small]
        # the plan patches every recorded pointer that falls inside an external's extent [data_ptr, data_ptr + numel * itemsize): only true
        # for contiguous tensors (cv: a row-padded view whose extent comes from its strides, _PlanRecord)
        for t in ext:
            if t is not None and t is not cv and not t.is_contiguous():
                return self.local_refiner("refiner", hidden, ctx, disp, conf, occ, cv, None, it, small=small, want_small=want_small)
        if ent == "warm":
            plan, pool, scratch = hip.Plan(), torch.cuda.MemPool(), {}
            outer = (self._bufs, self._ns)
            self._bufs, self._ns = scratch, ("plan", it)          # the iteration's persistent scratch belongs to the plan as well
            try:
                with torch.cuda.use_mem_pool(pool, device=self.device):
                    with _PlanRecord(plan, ext, cv):
                        res = self.local_refiner("refiner", hidden, ctx, disp, conf, occ, cv, None, it, small=small, want_small=want_small)
            finally:
                self._bufs, self._ns = outer
            store[key] = (plan, res, pool, scratch)
            return res
        plan, res, _, _ = ent
        plan.refine_step(*ext)
        return res

    # ---- upsampling masks ----------------------------------------------------------------------------
    def mask4x(self, p: str, hidden: Tensor, f2x: Tensor) -> Tensor:
        """UpsampleMask4x (submodules.py:96-115) -> logits (B,H,W,16), 9 used."""
        sx, cx = self.convT2(p + ".conv_x")
        a = self.cconv(sx, [hidden], shuffle2=cx)
        b = self.cconv(self.std(p + ".conv_y"), [f2x])
        y = self.cconv(self.std(p + ".conv_concat.0"), [a, b], act=hip.ACT_RELU)
        s2, c2 = self.convT2(p + ".conv_concat.2")
        return self.cconv(s2, [y], shuffle2=c2)

    def mask1x(self, p: str, rgb8: Tensor, f2x: Tensor) -> Tensor:
        """UpsampleMask1x (submodules.py:118-145) -> logits (B,H,W,16), 9 used.  rgb8: (B,H,W,8) with the x4-upsampled disparity in
        channel 0 (written by K7) and the normalised image in channels 1..3."""
        ab = self.cconv(self.merged(p + "|dispRgb", [(p + ".conv_disp.0", 0, 1.0, True), (p + ".conv_rgb.0", 1, 1.0, True)], 8),
                        [rgb8], act=hip.ACT_RELU)
        sc, cc = self.convT2(p + ".conv_ctx")
        c = self.cconv(sc, [f2x], shuffle2=cc)
        s0, s2 = self.std(p + ".conv_concat.0"), self.std(p + ".conv_concat.2", transposed=True)
        cin = ab.shape[-1] + c.shape[-1]
        if (self.use_k12_head and cin == 48 and s0[2] == 3 and s0[3] == 3 and not getattr(s0, "korder", 0) and s2[2] == 1 and s2[3] == 1
                and tuple(s0[0].shape) == (s0[4], 9 * cin) and tuple(s2[0].shape) == (s2[4], s0[4]) and s2[4] <= 32
                and self.narrow_ok(3, 3, 1, cin, s0[4])):
            # conv_concat.0 -> ReLU -> conv_concat.2 (1x1) as ONE K12 launch: the head's MFMAs read the 3x3 layer's accumulators as they are
            # (pack.head_frag); the 48-channel full-resolution tensor is never written
            hf = self._wfrag.get(("head", s2[0].data_ptr()))
            if hf is None:
                hf = self._wfrag[("head", s2[0].data_ptr())] = pack.head_frag(s2[0])
            return hip.conv_narrow([ab, c], self.wnarrow(s0, 9), s0[1], 3, 3, s0[4], act=hip.ACT_RELU, head=(hf, s2[1], s2[4]))
        y = self.cconv(s0, [ab, c], act=hip.ACT_RELU)
        return self.cconv(s2, [y])

    # ---- whole forward, in three stages (bench.py brackets the middle one, K1, with HIP events) ----------
    @torch.no_grad()
    def features(self, img0: Tensor, img1: Tensor, x8: Optional[Tensor] = None):
        """normalise -> CNN backbone -> feature pyramid -> multi-resolution transformer (s2m2.py:140-150).  ``x8``: the normalised 8-channel
        image tensor when the caller has already run hip.image_prep (GraphRunner: eagerly, straight from the caller's images)."""
        B = img0.shape[0]
        if x8 is None:
            x8 = hip.image_prep(img0, img1, self.dtype)                         # (2B,H,W,8): channels 1..3 = normalised RGB, 0 free
        p = "cnn_backbone"                                                      # CNNEncoder (submodules.py:63-93)
        c0, c2 = self._conv0(), self.std(p + ".conv0.2")
        if tuple(c0[0].shape) == (16, 8) and tuple(c2[0].shape) == (16, 16) and c0[2] == 1 and c2[2] == 1:
            st = self._packed.get("stem|fp32")                                  # conv0 = 1x1 - GELU - 1x1 per pixel on the VALU (K8)
            if st is None:
                st = self._packed["stem|fp32"] = (c0[0].float().contiguous(), c0[1], c2[0].float().contiguous(), c2[1])
            t = hip.stem_mlp(x8, *st)
        else:
            t = self.cconv(c0, [x8], act=hip.ACT_GELU)
            t = self.cconv(c2, [t])
        t = self.cconv(self.std(p + ".conv1_down.0"), [t], act=hip.ACT_GELU, stride=2)
        f2 = self.cconv(self.std(p + ".conv1_down.2"), [t])
        f2 = hip.groupnorm_nhwc(f2, 8, self.p[p + ".norm1.weight"], self.p[p + ".norm1.bias"])
        t = self.cconv(self.std(p + ".conv2.0"), [f2], act=hip.ACT_GELU)
        f2 = self.cconv(self.std(p + ".conv2.2"), [t], epi=hip.EPI_ADD, aux0=f2)
        f4 = self.cconv(self.std(p + ".conv2_down.0", frag=False), [f2], stride=2)
        py = self.unet("feat_pyramid", f4)
        z = py
        # DispInit's LayerNorm (submodules.py:165,216) is folded into the launch that writes feature_tr_4x -- the last K9 chain of the
        # last transformer -- as a second output; K1 then is the correlation alone (hip.corr).  A/B switch S2M2_FUSE_K1LN=0: K1 with
        # its own LayerNorm (hip.ln_corr), as for the widths whose chain has no LayerNorm output (C = 192, 384).
        self._tokens_normed = None
        for i in range(self.ntr):
            last = i == self.ntr - 1 and self.fuse_k1ln
            z = self.mrt(f"transformer.uformer_list.{i}", *z, ln_out=(self.ln_w, self.ln_b, 1e-5) if last else None)
        return z[0], py[0], f2[:B], x8[:B]                                      # tokens (2B,h,w,C), pyramid 1/4, left 1/2 features, image

    @torch.no_grad()
    def cost_volume(self, tr: Tensor, out: Optional[Tensor] = None, banded: bool = True, normed: Optional[Tensor] = None) -> Tensor:
        """K1: LayerNorm + all-pairs correlation (submodules.py:216-217); ``normed``: the tokens already normalised by the launch
        that produced ``tr`` (features()) -> the correlation alone.  The volume's rows start on 128-byte lines (hip.cv_alloc; a
        ``[..., :w]`` view of a padded allocation that K2 / K3 read with its pitch).  With ``k1_events`` set (bench.py) every launch
        carries a start / stop HIP event pair on its dispatch (hip.KernelTimer), collected in that list."""
        timer = None
        if self.k1_events is not None:
            timer = hip.KernelTimer()
            self.k1_events.append(timer)
        band = self.cv_band if banded else -1
        if normed is not None:
            if out is None:
                out = self.cv_buffer(tr)
            return hip.corr(normed, out=out, timer=timer, band=band)
        if out is None:
            out = self.cv_buffer(tr)
        return hip.ln_corr(tr, self.ln_w, self.ln_b, out=out, timer=timer, band=band)

    def cv_buffer(self, tr: Tensor) -> Tensor:
        twoB, h, w, _ = tr.shape
        return hip.cv_alloc(twoB // 2, h, w, tr.dtype, tr.device)

    @torch.no_grad()
    def finish(self, tr: Tensor, py0: Tensor, f2_left: Tensor, x8: Tensor, cv: Tensor, cap: Optional[dict] = None):
        """Sinkhorn + regression, global refiner, refinement loop, convex upsampling (s2m2.py:153-197)."""
        B = cv.shape[0]
        tr0 = tr[:B]
        # parity tests only ("teacher forcing", cap["inject"]): continue from the checker's tensors at a stage boundary, so that a
        # legitimate flip of a near-tie argmax is not what the stages downstream are judged on
        inj = (cap.get("inject") or {}) if cap is not None else {}
        if "cv" in inj:
            cv = inj["cv"].to(cv.device, cv.dtype).contiguous()
        disp, conf, occ, amax = hip.sinkhorn_regress(cv, self.use_positivity, 3, want_argmax=True)
        if cap is not None:
            cap.update(feature_tr_4x=tr.permute(0, 3, 1, 2), feature_py_4x=py0.permute(0, 3, 1, 2), cv=cv, argmax=amax,
                       disp0=disp, conf0=conf, occ0=occ)
            if "disp0" in inj:
                disp, conf, occ = (inj[k].to(cv.device, torch.float32).contiguous() for k in ("disp0", "conf0", "occ0"))
        disp = self.global_refiner("global_refiner", tr0, disp, conf)
        if cap is not None:
            cap["disp_g"] = disp
        # (measured and dropped, profiles/r04/ab_k1_store_sc1lib_overlap.txt: this branch -- and the f2x layers of the two mask heads -- on a second stream as
        # parallel branches of the captured hipGraph: 8.95 vs 8.83 ms per pair and K1 19.4 vs 17.7 us, the side kernels evict K1's tokens)
        fus = self.fusion("feat_fusion_layer", tr0, py0[:B])
        c0, c2 = self.std("ctx_feat.0"), self.std("ctx_feat.2")
        cc = fus.shape[-1]
        if (c0[2] == 1 and c2[2] == 1 and tuple(c0[0].shape) == (cc, cc) and tuple(c2[0].shape) == (cc, cc) and not getattr(c0, "korder", 0)
                and self.chain_frag_ok(cc)):
            # ctx_feat = 1x1 - GELU - 1x1 (s2m2.py:59,165): ONE two-stage K9 launch (the intermediate never leaves the CU) instead of two
            ctx = hip.mlp_chain(fus, [(self.wfrag(c0), c0[1], hip.ACT_GELU, None), (self.wfrag(c2), c2[1], hip.ACT_NONE, None)], frag=True)
        else:
            ctx = self.cconv(c2, [self.cconv(c0, [fus], act=hip.ACT_GELU)])
        hidden = hip.tanh(ctx)
        if cap is not None:
            cap["ctx"] = ctx.permute(0, 3, 1, 2)
        small = None
        for it in range(self.refine_iter):
            more = it + 1 < self.refine_iter                       # the epilogue also writes the next iteration's side input
            if self.native_refine and cap is None and hip.METER is None and disp.is_contiguous():
                res = self.refine_step_native(it, hidden, ctx, disp, conf, occ, cv, small, more)
            else:
                res = self.local_refiner("refiner", hidden, ctx, disp, conf, occ, cv, cap, it, small=small, want_small=more)
            hidden, disp, conf, occ = res[:4]
            small = res[4] if more else None
            if cap is not None:
                cap[f"disp_it{it}"], cap[f"conf_it{it}"], cap[f"occ_it{it}"] = disp, conf, occ
        m4 = self.mask4x("upsample_mask_4x_refine", hidden, f2_left)
        d_up, o_up, c_up = hip.convex_upsample([disp, occ, conf], m4, 4, scales=[4.0, 1.0, 1.0], chan_out=x8[..., 0])
        m1 = self.mask1x("upsample_mask_1x", x8, f2_left)
        if cap is not None:
            cap.update(hidden=hidden.permute(0, 3, 1, 2), mask4x=m4[..., :9].permute(0, 3, 1, 2), disp_up4=d_up,
                       mask1x=m1[..., :9].permute(0, 3, 1, 2))
        up = self.output_upsample
        return tuple(hip.convex_upsample([d_up, o_up, c_up], m1, 2 if up else 1, scales=[2.0 if up else 1.0, 1.0, 1.0], logit_up2=up))

    @torch.no_grad()
    def run(self, img0: Tensor, img1: Tensor, cap: Optional[dict] = None, x8: Optional[Tensor] = None):
        tr, py0, f2_left, x8 = self.features(img0, img1, x8)
        normed = self._tokens_normed
        if cap is not None and "feature_tr_4x" in (cap.get("inject") or {}):           # parity tests only, see finish()
            tr = cap["inject"]["feature_tr_4x"].to(tr.device, tr.dtype).permute(0, 2, 3, 1).contiguous()
            normed = self._normed_like_the_forward(tr) if normed is not None else None
        cv = self.cost_volume(tr, banded=cap is None, normed=normed)   # captured runs hand out the full volume, like the reference
        return self.finish(tr, py0, f2_left, x8, cv, cap)

    def _normed_like_the_forward(self, tr: Tensor):
        """Parity tests only (injected ``feature_tr_4x``).  In a free-running forward DispInit's LayerNorm is the second output of the K9
        launch that writes the tokens (attn_ffn) and K1 is the correlation alone (s2m2_corr) -- the kernel bench.py's roofline line
        measures.  So that injected runs exercise THAT pair and not K1's own-LayerNorm form, the injected tokens go through the same
        launch form: a one-stage K9 chain with an identity weight (x * 1 summed with zeros in fp32 is exact in both modes) whose
        LayerNorm output feeds hip.corr."""
        n, h, w, c = tr.shape
        eye = self._packed.get("identity|k9")
        if eye is None:
            eye = self._packed["identity|k9"] = Spec((torch.eye(c, device=tr.device, dtype=tr.dtype).contiguous(), None, 1, 1, c))
        grp = (h // 8) * w if h % 8 == 0 else 0
        ln_out = (self.ln_w, self.ln_b, 1e-5)
        if self.chain_frag_ok(c):
            return hip.mlp_chain(tr, [(self.wfrag(eye), None, hip.ACT_NONE, None)], ln_out=ln_out, xcd_group_rows=grp, frag=True)[1]
        return hip.mlp_chain(tr, [(eye[0], None, hip.ACT_NONE, None)], ln_out=ln_out, xcd_group_rows=grp)[1]

    def _conv0(self) -> Spec:
        """cnn_backbone.conv0.0 (1x1, 3 -> 16) reading the RGB planes from channels 1..3 of the 8-channel input tensor."""
        return self.merged("cnn_backbone.conv0.0|rgb@1", [("cnn_backbone.conv0.0", 1, 1.0, False)], 8)


class _PlanRecord:
    """hip.Plan.record with the cost volume's extent taken from its strides (a row-padded ``[..., :w]`` view: numel() undercounts it)"""

    def __init__(self, plan, ext, cv):
        self.plan, self.ext, self.cv = plan, ext, cv

    def __enter__(self):
        class _Span:                                               # duck-typed "tensor" carrying base pointer and byte extent
            def __init__(s, t, nbytes):
                s.t, s.nbytes = t, nbytes

            def data_ptr(s):
                return s.t.data_ptr()

            def numel(s):
                return s.nbytes

            def element_size(s):
                return 1
        cv = self.cv
        span = _Span(cv, cv.shape[0] * cv.stride(0) * cv.element_size())
        self.rec = self.plan.record([span if t is cv else t for t in self.ext])
        return self.rec.__enter__()

    def __exit__(self, *a):
        return self.rec.__exit__(*a)


class GraphRunner:
    """hipGraph replay of the forward for one (batch, height, width): the ~350 kernel launches of a forward cost more host time
    than GPU time once the kernels are fast, so they are captured once (``torch.cuda.graph`` = hipStreamBeginCapture on the stream
    every C-ABI call enqueues on) and replayed.  With ``split_k1`` the graph is cut around K1 so that bench.py can bracket that one
    kernel with HIP events inside the timed region: features graph -> K1 (eager) -> finish graph."""

    def __init__(self, eng: Engine, B: int, H: int, W: int, split_k1: bool = False):
        self.eng = eng
        self.split = split_k1
        dev = eng.device
        self.bufs: Dict[object, Tensor] = {}                      # this graph's persistent scratch (dies with the runner)
        outer = (eng._bufs, eng._ns)
        eng._bufs, eng._ns = self.bufs, "graph"
        try:
            self._build(eng, B, H, W, split_k1, dev)
        finally:
            eng._bufs, eng._ns = outer

    def _build(self, eng: "Engine", B: int, H: int, W: int, split_k1: bool, dev) -> None:
        self.l = torch.zeros((B, 3, H, W), device=dev, dtype=torch.float32)
        self.r = torch.zeros((B, 3, H, W), device=dev, dtype=torch.float32)
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(2):                                     # warm-up on the capture stream: packs weights, sizes every pool
                eng.run(self.l, self.r)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        # thread-local capture mode: with torch.distributed initialised, the RCCL watchdog thread polls events while this thread
        # captures; in the default (global) mode such a call from another thread invalidates the capture
        mode = "thread_local"
        # S2M2_EAGER_PREP (default 1): image_prep runs eagerly in front of the replay, straight from the caller's images into the graph's
        # static 8-channel tensor -- the two device-to-device copies of the images into static input buffers (2 x 15 MB at 1216 x 1024,
        # 14 us per forward) are gone; 0: the images are copied into static buffers and image_prep is the graph's first node
        self.x8 = None
        if os.environ.get("S2M2_EAGER_PREP", "1") != "0":
            self.x8 = hip.image_prep(self.l, self.r, eng.dtype)
        if not split_k1:
            self.g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g, capture_error_mode=mode):
                self.out = eng.run(self.l, self.r, x8=self.x8)
        else:
            self.ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.ga, capture_error_mode=mode):
                self.state = eng.features(self.l, self.r, self.x8)
            tr = self.state[0]
            self.normed = eng._tokens_normed
            self.cv = eng.cv_buffer(tr)
            eng.cost_volume(tr, out=self.cv, normed=self.normed)
            self.gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gb, pool=self.ga.pool(), capture_error_mode=mode):
                self.out = eng.finish(*self.state, self.cv)

    def __call__(self, img0: Tensor, img1: Tensor):
        if self.x8 is not None:
            hip.image_prep(img0, img1, self.eng.dtype, out=self.x8)
        else:
            self.l.copy_(img0)
            self.r.copy_(img1)
        if not self.split:
            self.g.replay()
        else:
            eng = self.eng
            self.ga.replay()
            eng.cost_volume(self.state[0], out=self.cv, normed=self.normed)   # eager between the two graphs: carries the timing events
            self.gb.replay()
        base = self.out[0]._base                                   # the three maps are slices of one allocation (hip.convex_upsample)
        if base is not None and all(o._base is base for o in self.out):
            b = base.clone()                                       # one copy out of the graph's static output buffer
            return tuple(b[k] for k in range(len(self.out)))
        return tuple(o.clone() for o in self.out)
