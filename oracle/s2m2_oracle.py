"""CPU oracle for the S2M2 inference hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module; the product (``s2m2_amd``) must never route through it.

What it is: a functional, fp32, CPU restatement of ``S2M2.forward`` of the reference
(/root/reference/src/s2m2/core/model/*.py, cited per function below), written as plain functions over
a flat ``state_dict`` (no ``nn.Module`` tree).  Dense layers use ``torch.nn.functional`` CPU ops; the
chain the product accelerates with hand-written kernels (LayerNorm + correlation, Sinkhorn optimal
transport, argmax + window regression, cost-volume lookups, convex upsampling, attention) is restated
with explicit arithmetic so every step of SURVEY.md Appendix A is visible.

Pinning: ``tests/golden/make_golden.py`` runs the *unmodified reference* in the build container on
seeded weights/inputs and stores its stage outputs in ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this oracle against them (the reference has no tests or golden
vectors of its own).
"""
from __future__ import annotations

from typing import Dict, List, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Mapping[str, Tensor]


# --------------------------------------------------------------------------------------------------
# precision mode.  "fp32" (default) is the parity configuration.  "fp16" EMULATES how the reference is deployed
# (fp32 weights + inputs under ``torch.amp.autocast(float16)``, model_utils.py:75-76) on the CPU: every tensor the CUDA
# autocast policy would hold in fp16 is rounded to fp16 at the op that produces it (``_q``) while the arithmetic inside an op
# stays fp32 (= fp16 operands, fp32 accumulation, one rounding of the result, which is what the GPU libraries do).
# Autocast policy used (PyTorch CUDA op lists): conv / conv_transpose / linear / matmul / einsum / SDPA
# -> fp16 in and out; layer_norm, group_norm, softmax, exp, log, sum, grid_sampler -> fp32 in and out; everything else
# runs in its widest input dtype (so fp16 (+,*,gelu,sigmoid,tanh,avg_pool,interpolate,logit) fp16 -> one fp16 rounding each).
# PARITY of this mode: bit-level UNPINNED -- the reference cannot run its CUDA autocast path in the CPU-only build container, and fp16
# runs that round at different points diverge by fp16 noise, so no golden can pin it bit for bit.  It is pinned STATISTICALLY against the
# reference's own fp16 run that does exist here (its run_stereo_matching autocast block with device cpu, tests/golden/make_golden_fp16.py):
# tests/test_fp16_reference_autocast.py checks that this mode is as close to those outputs as the reference's fp32 run is.  It is used
# only to QUANTIFY how far the HIP fp16 mode is from the reference's deployment numerics.  The fp32 mode (the parity configuration) IS
# pinned bit-tight: tests/test_oracle_golden.py.
# --------------------------------------------------------------------------------------------------
class _Prec:
    half = False


def _q(x: Tensor) -> Tensor:
    """fp16 storage point of the reference's autocast path (identity in fp32 mode)."""
    return x.half().float() if _Prec.half else x


def _gelu(x: Tensor) -> Tensor:
    return _q(F.gelu(x))


# --------------------------------------------------------------------------------------------------
# dense building blocks
# --------------------------------------------------------------------------------------------------
def _wb(sd: SD, p: str):
    w, b = sd[p + ".weight"], sd.get(p + ".bias")
    return (_q(w), _q(b) if b is not None else None)


def _conv(sd: SD, p: str, x: Tensor, stride: int = 1, pad=0) -> Tensor:
    w, b = _wb(sd, p)
    return _q(F.conv2d(_q(x), w, b, stride=stride, padding=pad))


def _convT(sd: SD, p: str, x: Tensor, stride: int = 1, pad: int = 0) -> Tensor:
    w, b = _wb(sd, p)
    return _q(F.conv_transpose2d(_q(x), w, b, stride=stride, padding=pad))


def _lin(sd: SD, p: str, x: Tensor) -> Tensor:
    w, b = _wb(sd, p)
    return _q(F.linear(_q(x), w, b))


def _ln(x: Tensor) -> Tensor:
    """Pre-norm LayerNorm without affine, eps 1e-5 (attentions.py:117,148,182,213,243)."""
    return F.layer_norm(x, (x.shape[-1],))


def _down(sd: SD, p: str, x: Tensor) -> Tensor:
    """AvgPool2d(2) then 1x1 conv (unet.py:25-30, stacked_MRT.py:22-27)."""
    return _conv(sd, p + ".1", _q(F.avg_pool2d(x, 2)))


def _up(sd: SD, p: str, x: Tensor) -> Tensor:
    """bilinear x2 (align_corners=False) then 1x1 conv (unet.py:32-37, stacked_MRT.py:29-34)."""
    return _conv(sd, p + ".1", _q(F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=False)))


def cnn_encoder(sd: SD, p: str, x: Tensor) -> Tuple[Tensor, Tensor]:
    """submodules.py:63-93."""
    x = _conv(sd, p + ".conv0.2", _gelu(_conv(sd, p + ".conv0.0", x)))
    y = _conv(sd, p + ".conv1_down.2", _gelu(_conv(sd, p + ".conv1_down.0", x, 2, 2)), 1, 1)
    y = F.group_norm(y, 8, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])          # autocast: fp32 out; the sum below stays fp32
    y = _conv(sd, p + ".conv2.2", _gelu(_conv(sd, p + ".conv2.0", y, 1, 1)), 1, 1) + y
    x4 = _conv(sd, p + ".conv2_down.0", y, 2, 1)
    return x4, y


def conv_block(sd: SD, p: str, z: Tensor) -> Tensor:
    """attentions.py:255-281: 3x3,GELU,3x3  +  1x1,ReLU,1x1."""
    a = _conv(sd, p + ".convs.2", _gelu(_conv(sd, p + ".convs.0", z, 1, 1)), 1, 1)
    b = _conv(sd, p + ".convs_1x.2", F.relu(_conv(sd, p + ".convs_1x.0", z)))
    return _q(a + b)


def feature_fusion(sd: SD, p: str, z0: Tensor, z1: Tensor) -> Tensor:
    """feature_fusion.py:24-31 (gate clamp [0.01, 0.99]); kernel size read from the weight."""
    z = torch.cat([z0, z1], 1)
    k = sd[p + ".feature_gate.0.weight"].shape[-1]
    g = _q(torch.sigmoid(_conv(sd, p + ".feature_gate.2", _gelu(_conv(sd, p + ".feature_gate.0", z, 1, k // 2)))))
    g = _q(g.clamp(0.01, 0.99))                                       # fp16 mode: the bounds land on fp16 values (0.01 -> 0.010002)
    f = _conv(sd, p + ".feature_fusion.2", _gelu(_conv(sd, p + ".feature_fusion.0", z, 1, k // 2)))
    return _q(_q(f + _q(g * z0)) + _q(_q(1 - g) * z1))


# --------------------------------------------------------------------------------------------------
# attention (attentions.py:8-96) -- explicit softmax(q k^T / sqrt(d)) v
# --------------------------------------------------------------------------------------------------
def _heads(x: Tensor, nh: int) -> Tensor:
    b, n, c = x.shape
    return x.reshape(b, n, nh, c // nh).permute(0, 2, 1, 3)


def _sdpa(q: Tensor, k: Tensor, v: Tensor, explicit: bool = False) -> Tuple[Tensor, Tensor]:
    """fp16 mode: ``explicit`` = the einsum / softmax / einsum spelling of the PE blocks (attentions.py:42-45: the scaled q and
    the score matrix are fp16 tensors, softmax returns fp32); otherwise F.scaled_dot_product_attention on fp16 q/k/v (flash:
    scores and softmax stay fp32 inside the kernel, the probabilities are rounded to fp16 for the PV product)."""
    d = q.shape[-1]
    if _Prec.half and not explicit:
        s = torch.matmul(q, k.transpose(-1, -2)) * (d ** -0.5)
    else:
        s = _q(torch.matmul(_q(q * (d ** -0.5)), k.transpose(-1, -2)))
    a = torch.softmax(s, dim=-1)
    return _q(torch.matmul(_q(a), v)), a


def self_attn(sd: SD, p: str, x: Tensor, nh: int, pe: Optional[Tensor]) -> Tensor:
    """attentions.py:33-54.  ``pe`` (N,N,32) only for blocks that own a ``pe_proj``."""
    b, n, c = x.shape
    q, k, v = (_heads(_lin(sd, p + "." + t, x), nh) for t in ("q", "k", "v"))
    use_pe = (p + ".pe_proj.weight") in sd
    o, a = _sdpa(q, k, v, explicit=use_pe)
    if use_pe:
        assert pe is not None
        pe_sum = _q(torch.einsum("bhij,ijc->bhic", _q(a), _q(pe)))           # attentions.py:47
        o = _q(o + _lin(sd, p + ".pe_proj", pe_sum))
    o = o.permute(0, 2, 1, 3).reshape(b, n, -1)
    return _lin(sd, p + ".proj", o)


def cross_attn(sd: SD, p: str, x: Tensor, y: Tensor, nh: int) -> Tuple[Tensor, Tensor]:
    """attentions.py:77-96: symmetric cross attention with shared q/k/v/proj weights."""
    b, n, c = x.shape
    qx, kx, vx = (_heads(_lin(sd, p + "." + t, x), nh) for t in ("q", "k", "v"))
    qy, ky, vy = (_heads(_lin(sd, p + "." + t, y), nh) for t in ("q", "k", "v"))
    ox, _ = _sdpa(qx, ky, vy)
    oy, _ = _sdpa(qy, kx, vx)
    ox = _lin(sd, p + ".proj", ox.permute(0, 2, 1, 3).reshape(b, n, -1))
    oy = _lin(sd, p + ".proj", oy.permute(0, 2, 1, 3).reshape(b, n, -1))
    return ox, oy


def ffn(sd: SD, p: str, z: Tensor) -> Tensor:
    """attentions.py:245-250."""
    return _q(_lin(sd, p + ".ffn.2", _gelu(_lin(sd, p + ".ffn.0", _ln(z)))) + z)


def _cross_block(sd: SD, p: str, z: Tensor, nh: int, two_d: bool) -> Tensor:
    """CrossAttnBlock1D/2D (attentions.py:150-161, 215-226); z: (2B,H,W,C), left = first half."""
    zn = _ln(z)
    x, y = zn.chunk(2, 0)
    b, h, w, c = x.shape
    shp = (b, h * w, c) if two_d else (b * h, w, c)
    ox, oy = cross_attn(sd, p + ".attn", x.reshape(shp), y.reshape(shp), nh)
    return _q(torch.cat([ox.reshape(b, h, w, c), oy.reshape(b, h, w, c)], 0) + z)


def _self_block(sd: SD, p: str, z: Tensor, nh: int, two_d: bool, pe: Optional[Tensor]) -> Tensor:
    """SelfAttnBlock1D/2D (attentions.py:119-128, 185-193)."""
    b, h, w, c = z.shape
    shp = (b, h * w, c) if two_d else (b * h, w, c)
    zz = z.reshape(shp)
    return _q(self_attn(sd, p + ".attn", _ln(zz), nh, pe) + zz).reshape(b, h, w, c)


def global_attn_block(sd: SD, p: str, z: Tensor, nh: int, pe: Optional[Tensor]) -> Tensor:
    """attentions.py:311-321, NCHW in/out; cross part present iff its weights are."""
    z = z.permute(0, 2, 3, 1)
    if (p + ".cross_attn.attn.q.weight") in sd:
        z = ffn(sd, p + ".ffn_c", _cross_block(sd, p + ".cross_attn", z, nh, True))
    z = ffn(sd, p + ".ffn", _self_block(sd, p + ".self_attn", z, nh, True, pe))
    return z.permute(0, 3, 1, 2).contiguous()


def basic_attn_block(sd: SD, p: str, z: Tensor, nh: int) -> Tensor:
    """attentions.py:347-355 (1-D attention along image rows)."""
    z = z.permute(0, 2, 3, 1)
    z = ffn(sd, p + ".ffn_c", _cross_block(sd, p + ".cross_attn", z, nh, False))
    z = ffn(sd, p + ".ffn", _self_block(sd, p + ".self_attn", z, nh, False, None))
    return z.permute(0, 3, 1, 2)


def _count(sd: SD, prefix: str) -> int:
    n = 0
    while any(k.startswith(f"{prefix}.{n}.") for k in sd):
        n += 1
    return n


# --------------------------------------------------------------------------------------------------
# positional encoding (utils.py:32-60) -- separable tables, then the dense (N,N,32) tensor
# --------------------------------------------------------------------------------------------------
def pe_tables(h: int, w: int, pe_dim: int = 32) -> Tuple[Tensor, Tensor]:
    """(2w+1, 16) and (2h+1, 16) L2-normalised sinc tables; pe[i,j] = .5*[px[dx+w-1], py[dy+h-1]]."""
    def table(n: int) -> Tensor:
        L = 2 * n + 1
        sig = 5 / pe_dim
        pos = torch.linspace(-3, 3, L).tanh()
        dim_t = torch.linspace(-1, 1, pe_dim // 2)
        x = (dim_t[None, :] - pos[:, None]) / sig
        s = torch.where(x.abs() < 1e-6, torch.ones_like(x), torch.sin(3.1415 * x) / (3.1415 * x))
        return F.normalize(s, p=2, dim=-1)
    return table(w), table(h)


def dense_pe(h: int, w: int, pe_dim: int = 32) -> Tensor:
    px, py = pe_tables(h, w, pe_dim)
    ys, xs = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    xs, ys = xs.reshape(-1), ys.reshape(-1)
    rel_x = xs[:, None] - xs[None, :] + w - 1
    rel_y = ys[:, None] - ys[None, :] + h - 1
    return 0.5 * torch.cat([px[rel_x], py[rel_y]], dim=2)


# --------------------------------------------------------------------------------------------------
# U-Net (unet.py:65-112) and multi-resolution transformer (stacked_MRT.py:89-133)
# --------------------------------------------------------------------------------------------------
def unet(sd: SD, p: str, z: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    use_pe = (p + ".enc3s.0.self_attn.attn.pe_proj.weight") in sd
    pe = dense_pe(z.shape[-2] // 8, z.shape[-1] // 8) if use_pe else None
    z0 = conv_block(sd, p + ".enc0", z)
    z1 = conv_block(sd, p + ".enc1", _down(sd, p + ".down_conv0", z0))
    z2 = conv_block(sd, p + ".enc2", _down(sd, p + ".down_conv1", z1))
    z3 = _down(sd, p + ".down_conv2", z2)
    for i in range(_count(sd, p + ".enc3s")):
        z3 = global_attn_block(sd, f"{p}.enc3s.{i}", z3, 8, pe)
    for i in range(_count(sd, p + ".dec3s")):
        z3 = global_attn_block(sd, f"{p}.dec3s.{i}", z3, 8, None)
    n2 = conv_block(sd, p + ".dec2", feature_fusion(sd, p + ".concat_conv2", z2, _up(sd, p + ".up_conv2", z3)))
    n1 = conv_block(sd, p + ".dec1", feature_fusion(sd, p + ".concat_conv1", z1, _up(sd, p + ".up_conv1", n2)))
    n0 = conv_block(sd, p + ".dec0", feature_fusion(sd, p + ".concat_conv0", z0, _up(sd, p + ".up_conv0", n1)))
    return n0, n1, n2, z3


def mrt(sd: SD, p: str, z0: Tensor, z1: Tensor, z2: Tensor, z3: Tensor, nh: int = 1):
    z0 = basic_attn_block(sd, p + ".enc_attn0", z0, nh)
    z1 = feature_fusion(sd, p + ".down_concat1", z1, _down(sd, p + ".down_conv0", z0))
    z1 = basic_attn_block(sd, p + ".enc_attn1", z1, 2 * nh)
    z2 = feature_fusion(sd, p + ".down_concat2", z2, _down(sd, p + ".down_conv1", z1))
    z2 = basic_attn_block(sd, p + ".enc_attn2", z2, 4 * nh)
    z3 = feature_fusion(sd, p + ".down_concat3", z3, _down(sd, p + ".down_conv2", z2))
    for i in range(2):
        z3 = global_attn_block(sd, f"{p}.enc_attn3s.{i}", z3, 8 * nh, None)
    for i in range(2):
        z3 = global_attn_block(sd, f"{p}.dec_attn3s.{i}", z3, 8 * nh, None)
    z2 = basic_attn_block(sd, p + ".dec_attn2", feature_fusion(sd, p + ".up_concat2", z2, _up(sd, p + ".up_conv2", z3)), 4 * nh)
    z1 = basic_attn_block(sd, p + ".dec_attn1", feature_fusion(sd, p + ".up_concat1", z1, _up(sd, p + ".up_conv1", z2)), 2 * nh)
    z0 = basic_attn_block(sd, p + ".dec_attn0", feature_fusion(sd, p + ".up_concat0", z0, _up(sd, p + ".up_conv0", z1)), nh)
    return z0, z1, z2, z3


# --------------------------------------------------------------------------------------------------
# DispInit: LayerNorm + all-pairs correlation + Sinkhorn OT + argmax + window regression
# (submodules.py:154-243)
# --------------------------------------------------------------------------------------------------
def ln_corr(feat: Tensor, gamma: Tensor, beta: Tensor) -> Tensor:
    """feat (2B,C,h,w) NCHW, left = first B.  cv[b,y,i,j] = <LN(f0[b,:,y,i]), LN(f1[b,:,y,j])>."""
    f = F.layer_norm(feat.permute(0, 2, 3, 1), (feat.shape[1],), gamma, beta, 1e-5)
    f0, f1 = f.chunk(2, 0)
    return _q(torch.matmul(_q(f0), _q(f1).transpose(-1, -2)))          # (B,h,w,w), no 1/sqrt(C); fp16 mode: einsum in/out fp16


def _lse(x: Tensor, dim: int, x_is_half: bool = False) -> Tensor:
    """logsumexp_stable (submodules.py:147-152): m + log(max(sum exp(x-m), 1e-30)).  fp16 mode: only the first call sees an
    fp16 tensor (the padded cost volume), so only there is ``x - m`` an fp16 result; exp/sum/log are fp32 ops under autocast
    and every later argument (cv + fp32 potentials) is fp32 by type promotion."""
    m = x.amax(dim=dim, keepdim=True)
    xm = _q(x - m) if x_is_half else x - m
    s = xm.exp().sum(dim=dim, keepdim=True).clamp_min(1e-30)
    return (m + s.log()).squeeze(dim)


def sinkhorn_prob(cv: Tensor, use_positivity: bool, ot_iter: int = 3) -> Tensor:
    """Masked transport probabilities Pm (B,h,w,w); Appendix A steps 3-7."""
    w = cv.shape[-1]
    S = cv
    if use_positivity:
        upper = torch.ones(w, w, dtype=torch.bool).triu(1)             # j > i
        S = S.masked_fill(upper, -1e4)
    S = F.pad(S, (0, 1, 0, 1))                                         # dustbin row + col = 0
    marg = torch.cat([torch.ones(w), torch.tensor([float(w)])]) / (2 * w)
    lmu = marg.log()
    lnu = marg.log()
    v = lnu - _lse(S, 2, x_is_half=_Prec.half)                         # over i
    u = lmu - _lse(S + v[:, :, None, :], 3)                            # over j
    for _ in range(ot_iter - 1):
        v = lnu - _lse(S + u[:, :, :, None], 2)
        u = lmu - _lse(S + v[:, :, None, :], 3)
    logp = S + u[:, :, :, None] + v[:, :, None, :]
    P = _q((logp[:, :, :-1, :-1] + torch.log(torch.tensor(2.0 * w))).exp())      # undo the 1/(2w) marginals; ``.to(dtype)`` :200
    if use_positivity:
        P = P.masked_fill(upper, 0)
    return P


def regress(P: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """argmax (first max wins) + 5-tap window expectation; Appendix A steps 8-10.

    Returns disp, conf, occ (B,1,h,w) and the int64 argmax index (B,h,w)."""
    B, h, w, _ = P.shape
    ind = P.argmax(dim=3)
    Pp = F.pad(P, (2, 2))
    conf = torch.zeros(B, h, w)
    num = torch.zeros(B, h, w)
    for k in range(-2, 3):
        pk = torch.gather(Pp, 3, (ind + k + 2)[..., None])[..., 0]
        conf = _q(conf + pk)                                           # fp16 mode: fp16 accumulators (submodules.py:232-236)
        num = _q(num + _q(pk * (ind + k)))
    corr = _q(_q(num + 1e-4) / _q(conf + 1e-4))
    xs = torch.linspace(0, w - 1, w)
    disp = _q(xs.reshape(1, 1, w) - corr)
    occ = P.sum(dim=3)                                                 # ``sum`` is an fp32 op under autocast
    return disp[:, None], conf[:, None], occ[:, None], ind


def disp_init(sd: SD, feat: Tensor, use_positivity: bool):
    cv = ln_corr(feat, sd["disp_init.layer_norm.weight"], sd["disp_init.layer_norm.bias"])
    P = sinkhorn_prob(cv, use_positivity)
    disp, conf, occ, ind = regress(P)
    return disp, conf, occ, cv, ind, P


# --------------------------------------------------------------------------------------------------
# cost-volume lookup (submodules.py:7-60; Appendix A step 11)
# --------------------------------------------------------------------------------------------------
def _sample_rows(img: Tensor, x: Tensor, yrow: Tensor) -> Tensor:
    """Bilinear sample with zeros padding, align_corners=True, restating the reference call chain.

    img (R, Hs, Ws): R independent single-channel images (one per (b, y) image row; Hs = left column i,
    Ws = right column j or j/2).  x (R, Hs, T) pixel x-coords, yrow (R, Hs, T) pixel y-coords.
    The reference converts pixel -> normalised coords in Python (``2*x/(W-1)-1``, submodules.py:12-13)
    and PyTorch's CPU kernel converts back with ``(g+1)*((size-1)/2)`` -- both fp32 -- so the effective
    coordinate differs from x by an ulp or so; that round trip is reproduced here on both axes.
    """
    R, Hs, Ws = img.shape
    Wt = torch.tensor(float(Ws))
    Ht = torch.tensor(float(Hs))
    if _Prec.half:      # bilinear_sampler's coordinate arithmetic runs on fp16 tensors; grid_sample itself is an fp32 op (autocast)
        gx = _q(_q(2 * x / (Wt - 1)) - 1)
        gy = _q(_q(2 * yrow / (Ht - 1)) - 1)
    else:
        gx = 2 * x / (Wt - 1) - 1
        gy = 2 * yrow / (Ht - 1) - 1
    ix = (gx + 1) * ((Ws - 1) / 2)
    iy = (gy + 1) * ((Hs - 1) / 2)
    x0 = ix.floor()
    y0 = iy.floor()
    wx = ix - x0
    wy = iy - y0
    ex = 1 - wx
    ey = 1 - wy
    out = torch.zeros_like(ix)
    flat = img.reshape(R, Hs * Ws)
    for (yy, xx, wt) in ((y0, x0, ey * ex), (y0, x0 + 1, ey * wx), (y0 + 1, x0, wy * ex), (y0 + 1, x0 + 1, wy * wx)):
        ok = (xx >= 0) & (xx <= Ws - 1) & (yy >= 0) & (yy <= Hs - 1)
        idx = (yy.clamp(0, Hs - 1) * Ws + xx.clamp(0, Ws - 1)).long()
        val = torch.gather(flat, 1, idx.reshape(R, -1)).reshape(ix.shape)
        out = out + torch.where(ok, val, torch.zeros_like(val)) * wt
    return out


def cv_lookup(cv: Tensor, disp: Tensor, radius: int = 4) -> Tuple[Tensor, Tensor]:
    """cv (B,h,w,w), disp (B,1,h,w) -> corr1, corr2 (B,2r+1,h,w): level 0 and the j-avg-pooled level 1."""
    B, h, w, _ = cv.shape
    img0 = cv.reshape(B * h, w, w)
    wh = w // 2
    img1 = _q((cv[..., 0:2 * wh:2] + cv[..., 1:2 * wh:2]) * 0.5).reshape(B * h, w, wh)    # avg-pool along j
    dx = torch.linspace(-radius, radius, 2 * radius + 1).reshape(1, 1, -1)
    i = torch.arange(w, dtype=torch.float32).reshape(1, w, 1)
    d = disp.reshape(B * h, w, 1)
    yrow = i + 0 * dx
    c1 = _sample_rows(img0, _q(_q(i - d) + dx), yrow.expand(B * h, w, -1))            # fp16 mode: coords, disp, dx are fp16 tensors
    c2 = _sample_rows(img1, _q(_q(i / 2 - _q(d / 2)) + dx), yrow.expand(B * h, w, -1))
    T = 2 * radius + 1
    return (c1.reshape(B, h, w, T).permute(0, 3, 1, 2).contiguous(),
            c2.reshape(B, h, w, T).permute(0, 3, 1, 2).contiguous())


# --------------------------------------------------------------------------------------------------
# refiners (refinenet.py)
# --------------------------------------------------------------------------------------------------
def _logit(x: Tensor, eps: float) -> Tensor:
    x = x.clamp(eps, 1 - eps)
    return torch.log(x / (1 - x))


def global_refiner(sd: SD, p: str, ctx: Tensor, disp: Tensor, conf: Tensor) -> Tensor:
    """refinenet.py:61-73."""
    mask = (conf > 0.2).float()
    x = torch.cat([_q(disp / 1e2) * mask, torch.logit(mask * conf, eps=1e-1), ctx], 1)
    f = _conv(sd, p + ".init_feat.2", _gelu(_conv(sd, p + ".init_feat.0", x, 1, 1)))
    f = unet(sd, p + ".refine_unet", f)[0]
    upd = _q(_conv(sd, p + ".out_feat.0", f, 1, 1) * 1e2)
    return _q(mask * disp + (1 - mask) * upd)


def conv_gru(sd: SD, p: str, h: Tensor, x: Tensor) -> Tensor:
    """refinenet.py:22-36: vertical (3x1) then horizontal (1x3) GRU pass."""
    for sfx, pad in (("1", (1, 0)), ("2", (0, 1))):
        hx = torch.cat([h, x], 1)
        z = _q(torch.sigmoid(_conv(sd, f"{p}.convz{sfx}", hx, 1, pad)))
        r = _q(torch.sigmoid(_conv(sd, f"{p}.convr{sfx}", hx, 1, pad)))
        q = _q(torch.tanh(_conv(sd, f"{p}.convq{sfx}", torch.cat([_q(r * h), x], 1), 1, pad)))
        h = _q(_q(_q(1 - z) * h) + _q(z * q))
    return h


def local_refiner(sd: SD, p: str, hidden: Tensor, ctx: Tensor, disp: Tensor, conf: Tensor, occ: Tensor, cv: Tensor,
                  occ_is_fp32: bool = False):
    """refinenet.py:126-154.  fp16 mode: ``occ`` arrives as an fp32 tensor on the first iteration (DispInit's ``sum``) and as
    fp16 afterwards (``.to(disp.dtype)``), which decides where its logit and the updated logit are rounded."""
    cl = _q(torch.logit(conf, eps=1e-2))
    ol = torch.logit(occ, eps=1e-2)
    if not occ_is_fp32:
        ol = _q(ol)
    c1, c2 = cv_lookup(cv, disp)
    f1 = _conv(sd, p + ".corr_feat1.2", _gelu(_conv(sd, p + ".corr_feat1.0", c1 / 16)))
    f2 = _conv(sd, p + ".corr_feat2.2", _gelu(_conv(sd, p + ".corr_feat2.0", c2 / 16)))
    fd = _conv(sd, p + ".disp_feat.2", _gelu(_conv(sd, p + ".disp_feat.0", _q(disp / 1e2), 1, 1)), 1, 1)
    fc = _conv(sd, p + ".conf_occ_feat.2", _gelu(_conv(sd, p + ".conf_occ_feat.0", torch.cat([cl, ol], 1), 1, 1)))
    x = torch.cat([fd, f1, f2, ctx, fc], 1)
    x = _conv(sd, p + ".disp_corr_ctx_cat.2", _gelu(_conv(sd, p + ".disp_corr_ctx_cat.0", x)), 1, 1)
    x = unet(sd, p + ".refine_unet", x)[0]
    hn = conv_gru(sd, p + ".gru", hidden, x)
    dd = _conv(sd, p + ".disp_update.2", _gelu(_conv(sd, p + ".disp_update.0", hn, 1, 1)), 1, 1)
    co = _conv(sd, p + ".conf_occ_update.2", _gelu(_conv(sd, p + ".conf_occ_update.0", hn, 1, 1)), 1, 1)
    so = co[:, 1:2] + ol
    if not occ_is_fp32:
        so = _q(so)
    return hn, _q(disp + dd), _q(torch.sigmoid(_q(co[:, 0:1] + cl))), _q(torch.sigmoid(so)), (c1, c2)


# --------------------------------------------------------------------------------------------------
# convex upsampling (s2m2.py:101-133, utils.py:9-20; Appendix A step 13)
# --------------------------------------------------------------------------------------------------
def _neigh9(x: Tensor) -> Tensor:
    """(B,1,h,w) -> (B,9,h,w): replicate-padded 3x3 neighbourhood, row-major (dy,dx)."""
    B, _, h, w = x.shape
    xp = F.pad(x, (1, 1, 1, 1), mode="replicate")
    return torch.cat([xp[:, :, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], 1)


def upsample4x(x: Tensor, logits: Tensor) -> Tensor:
    n = _neigh9(x)
    n = n.repeat_interleave(4, dim=2).repeat_interleave(4, dim=3)      # nearest x4
    return (n * logits.softmax(1)).sum(1, keepdim=True)


def upsample1x(x: Tensor, logits: Tensor, output_upsample: bool = False) -> Tensor:
    n = _neigh9(x)
    if output_upsample:
        n = n.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
        logits = _q(F.interpolate(logits, scale_factor=2, mode="bilinear", align_corners=False))
    return (n * logits.softmax(1)).sum(1, keepdim=True)


def upsample_mask_4x(sd: SD, p: str, hidden: Tensor, f2x: Tensor) -> Tensor:
    """submodules.py:110-115."""
    a = _convT(sd, p + ".conv_x", hidden, 2)
    b = _conv(sd, p + ".conv_y", f2x, 1, 1)
    y = F.relu(_conv(sd, p + ".conv_concat.0", torch.cat([a, b], 1), 1, 1))
    return _convT(sd, p + ".conv_concat.2", y, 2)


def upsample_mask_1x(sd: SD, p: str, disp: Tensor, rgb: Tensor, f2x: Tensor) -> Tensor:
    """submodules.py:137-145."""
    a = F.relu(_convT(sd, p + ".conv_disp.0", disp, 1, 1))
    b = F.relu(_convT(sd, p + ".conv_rgb.0", rgb, 1, 1))
    c = _convT(sd, p + ".conv_ctx", f2x, 2)
    y = F.relu(_conv(sd, p + ".conv_concat.0", torch.cat([a, b, c], 1), 1, 1))
    return _convT(sd, p + ".conv_concat.2", y)


# --------------------------------------------------------------------------------------------------
# whole forward (s2m2.py:136-197)
# --------------------------------------------------------------------------------------------------
@torch.no_grad()
def forward(sd: SD, img0: Tensor, img1: Tensor, use_positivity: bool = False, refine_iter: int = 3,
            output_upsample: bool = False, capture: Optional[Dict[str, Tensor]] = None, precision: str = "fp32",
            inject: Optional[Dict[str, Tensor]] = None):
    """Returns (disp_up, occ_up, conf_up), each (B,1,H,W) fp32.  ``capture`` (optional dict) receives the
    stage boundaries the parity tests compare: feature_tr_4x, cv, argmax, disp0/conf0/occ0, disp_g,
    per-iteration disp/conf/occ/corr, masks.  ``precision``: "fp32" (parity configuration) or "fp16" (emulation of the
    reference's autocast deployment mode, see the precision note at the top of this file).  ``inject`` (parity tests only):
    {"feature_tr_4x": (2B,C,h,w)} replaces the transformer output before DispInit, like the product's Engine ``inject`` hook -- used to
    run both sides from synthetic features with sharp, unambiguous matches."""
    if precision not in ("fp32", "fp16"):
        raise ValueError(precision)
    old = _Prec.half
    _Prec.half = precision == "fp16"
    try:
        return _forward(sd, img0, img1, use_positivity, refine_iter, output_upsample, capture, inject)
    finally:
        _Prec.half = old


def _forward(sd: SD, img0: Tensor, img1: Tensor, use_positivity: bool, refine_iter: int, output_upsample: bool,
             capture: Optional[Dict[str, Tensor]], inject: Optional[Dict[str, Tensor]] = None):
    cap = capture if capture is not None else {}
    sd = {k: v.float() for k, v in sd.items()}
    a = (img0.float() / 255.0 - 0.5) * 2
    b = (img1.float() / 255.0 - 0.5) * 2
    B = a.shape[0]
    f4, f2 = cnn_encoder(sd, "cnn_backbone", torch.cat([a, b], 0))
    f2_left = f2[:B]
    cap["feature_4x"] = f4
    py = unet(sd, "feat_pyramid", f4)
    cap["feature_py_4x"] = py[0]
    if inject and "feature_tr_4x" in inject:                  # (the transformers' output is replaced: they are not run)
        tr = _q(inject["feature_tr_4x"].float()).contiguous()
    else:
        z = py
        for i in range(_count(sd, "transformer.uformer_list")):
            z = mrt(sd, f"transformer.uformer_list.{i}", *z)
        tr = z[0].contiguous()
    cap["feature_tr_4x"] = tr
    disp, conf, occ, cv, ind, P = disp_init(sd, tr, use_positivity)
    cap.update(cv=cv, argmax=ind, disp0=disp, conf0=conf, occ0=occ, prob=P)
    tr0 = tr[:B].contiguous()
    disp = global_refiner(sd, "global_refiner", tr0, disp, conf)
    if use_positivity:
        disp = disp.clamp(min=0)
    cap["disp_g"] = disp
    fus = feature_fusion(sd, "feat_fusion_layer", tr0, py[0][:B])
    ctx = _conv(sd, "ctx_feat.2", _gelu(_conv(sd, "ctx_feat.0", fus)))
    hidden = _q(torch.tanh(ctx))
    cap["ctx"] = ctx
    w = disp.shape[-1]
    xs = torch.arange(w, dtype=torch.float32).reshape(1, 1, 1, w)
    for it in range(refine_iter):
        hidden, disp, conf, occ, corr = local_refiner(sd, "refiner", hidden, ctx, disp, conf, occ, cv,
                                                      occ_is_fp32=_Prec.half and it == 0)
        if use_positivity:
            disp = disp.clamp(min=0)
        occ = occ * (xs - disp >= 0)
        cap[f"disp_it{it}"] = disp
        cap[f"conf_it{it}"] = conf
        cap[f"occ_it{it}"] = occ
        cap[f"corr1_it{it}"], cap[f"corr2_it{it}"] = corr
    cap["hidden"] = hidden
    m4 = upsample_mask_4x(sd, "upsample_mask_4x_refine", hidden, f2_left)
    cap["mask4x"] = m4
    d_up = upsample4x(_q(disp * 4), m4)
    o_up = upsample4x(occ, m4)
    c_up = upsample4x(conf, m4)
    cap["disp_up4"] = d_up
    m1 = upsample_mask_1x(sd, "upsample_mask_1x", d_up, a, f2_left)
    cap["mask1x"] = m1
    d_up = upsample1x(d_up, m1, output_upsample)
    o_up = upsample1x(o_up, m1, output_upsample)
    c_up = upsample1x(c_up, m1, output_upsample)
    if output_upsample:
        d_up = 2 * d_up
    return d_up, o_up, c_up
