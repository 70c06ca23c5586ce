"""CPU stand-ins for the functions of :mod:`s2m2_amd.hip`, used ONLY by tests/test_engine_wiring_cpu.py to exercise the engine's
host logic (weight packing, channel padding, merged layers, stage wiring) without a GPU.  Each function restates the
documented semantics of its C-ABI entry point (include/s2m2_hip.h) with plain PyTorch CPU ops / the oracle."""
import types

import torch
import torch.nn.functional as F

from oracle import s2m2_oracle as O

ACTS = [lambda t: t, F.gelu, F.relu, torch.sigmoid, torch.tanh]


def row_attn_reference(x, heads, cross, weights, vectors, ln_eps=1e-5, ln_out_eps=None, rounding=None):
    """s2m2_row_attn (include/s2m2_hip.h) in plain PyTorch: z' = z + proj(attention(LN(z), LN(s)));  out = z' + ffn(LN(z')), s = the same line of
    image (n + nimg/2) % nimg (cross) or z.  weights (768, 128) in the row_attn packing, vectors (12, 128) as the entry point takes them.
    rounding: dtype every intermediate the kernel rounds is rounded to (None: fp32 throughout)."""
    from s2m2_amd.pack import rowattn_unpack
    rd = (lambda t: t.to(rounding).float()) if rounding is not None else (lambda t: t)
    wq, wk, wv, wp, w0, w2 = rowattn_unpack(weights).float().chunk(6, 0)
    v = vectors.float()
    bq, wsq, bk, wsk, bv, wsv, bp, b0, ws0, b2, g, b = (v[i] for i in range(12))
    for ws, w in ((wsq, wq), (wsk, wk), (wsv, wv), (ws0, w0)):
        assert torch.allclose(ws, w.sum(1), atol=2e-3)
    z = x.float()
    nimg, h, w, C = z.shape
    d = C // heads
    ln = lambda t: F.layer_norm(t, (C,), eps=ln_eps)                      # noqa: E731
    src = z.roll(nimg // 2, 0) if cross else z
    q, k, vv = rd(F.linear(ln(z), wq, bq)), rd(F.linear(ln(src), wk, bk)), rd(F.linear(ln(src), wv, bv))
    sp = lambda t: t.reshape(nimg * h, w, heads, d).transpose(1, 2)       # noqa: E731
    a = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) * d ** -0.5, -1)
    o = rd((a @ sp(vv)).transpose(1, 2).reshape(nimg, h, w, C))
    z1 = rd(rd(F.linear(o, wp, bp)) + z)
    hdn = rd(F.gelu(F.linear(ln(z1), w0, b0)))
    out = rd(rd(F.linear(hdn, w2, b2)) + z1).to(x.dtype)
    if ln_out_eps is None:
        return out
    return out, F.layer_norm(out.float(), (C,), g, b, ln_out_eps).to(x.dtype)


def make():
    ns = types.SimpleNamespace(ACT_NONE=0, ACT_GELU=1, ACT_RELU=2, ACT_SIGMOID=3, ACT_TANH=4,
                               EPI_NONE=0, EPI_ADD=1, EPI_MUL=2, EPI_GRU=3, EPI_GATEMIX=4, EPI_DUALMIX=5, load=lambda: None)

    def ln_corr(tokens, g, b, cv_dtype=None, out=None, timer=None, band=-1):
        return O.ln_corr(tokens.permute(0, 3, 1, 2).float(), g, b)

    def corr(tokens, cv_dtype=None, out=None, timer=None, band=-1):                  # tokens already normalised (K9's second output)
        t = tokens.float()
        B = t.shape[0] // 2
        cv = torch.einsum("bhic,bhjc->bhij", t[:B], t[B:])
        if out is not None:
            out.copy_(cv)
            return out
        return cv

    def cv_alloc(B, h, w, dtype, device, aligned=True):
        pitch = (w + 31) // 32 * 32 if aligned else w
        return torch.empty((B, h, w, pitch), dtype=dtype, device=device)[..., :w]     # the row-padded view the kernels see

    def sinkhorn_regress(cv, pos, ot_iter=3, want_argmax=False):
        d, c, o, ind = O.regress(O.sinkhorn_prob(cv.float(), pos, ot_iter))
        return (d, c, o, ind.int()) if want_argmax else (d, c, o)

    def cv_lookup_into(cv, disp, buf, off1, off2, radius=4):
        c1, c2 = O.cv_lookup(cv.float(), disp.float(), radius)
        T = 2 * radius + 1
        buf[..., off1:off1 + T] = c1.permute(0, 2, 3, 1).to(buf.dtype)
        buf[..., off2:off2 + T] = c2.permute(0, 2, 3, 1).to(buf.dtype)

    def conv2d(srcs, weight, bias, KH, KW, Cout, act=0, epi=0, aux0=None, aux1=None, out=None, out_scale=1.0, shuffle2=0, tile=0,
               stride=1, ln_wsum=None, ln_eps=1e-5, ksplit=0, bias2=None, pool2=False):
        if isinstance(srcs, torch.Tensor):
            srcs = [srcs]
        x = torch.cat([s.float() for s in srcs], -1)
        if pool2:
            assert KH == 1 and KW == 1 and stride == 1
            x = F.avg_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1).to(srcs[0].dtype).float()
        n, h, w, cin = x.shape
        if epi == 5:                                                 # DUALMIX: two 1x1 layers over consecutive channel ranges
            assert KH == 1 and KW == 1 and act == 3 and 0 < ksplit < cin
            wf = weight.float()
            g = torch.sigmoid(F.linear(x[..., :ksplit], wf[:, :ksplit], bias)).clamp(0.01, 0.99)
            y = F.linear(x[..., ksplit:], wf[:, ksplit:], bias2) + g * aux0.float() + (1 - g) * aux1.float()
            return y.to(srcs[0].dtype).contiguous()
        if ln_wsum is not None:
            assert KH == 1 and KW == 1 and torch.allclose(ln_wsum, weight.float().sum(1), atol=1e-4)
            x = F.layer_norm(x, (cin,), eps=ln_eps)
        assert tuple(weight.shape) == (Cout, KH * KW * cin) and Cout % 8 == 0 and cin % 8 == 0
        wt = weight.float().reshape(Cout, KH, KW, cin).permute(0, 3, 1, 2)
        y = F.conv2d(x.permute(0, 3, 1, 2), wt, bias, stride=stride, padding=(KH // 2, KW // 2))
        y = (ACTS[act](y) * out_scale).permute(0, 2, 3, 1)
        if epi == 1:
            y = y + aux0.float()
        elif epi == 2:
            y = y * aux0.float()
        elif epi == 3:
            y = (1 - aux0.float()) * aux1.float() + aux0.float() * y
        elif epi == 4:
            g = y.clamp(0.01, 0.99)
            y = g * aux0.float() + (1 - g) * aux1.float()
        if shuffle2:
            cp = shuffle2
            y = y.reshape(n, h, w, 2, 2, cp).permute(0, 1, 3, 2, 4, 5).reshape(n, 2 * h, 2 * w, cp)
        y = y.to(srcs[0].dtype)
        if out is not None:
            out.copy_(y)
            return out
        return y.contiguous()

    def mlp_chain_supported(C, dtype):
        return C in (128, 256)

    def mlp_chain_ln_out_supported(C, dtype):
        return C in (128, 256)

    def mlp_fan_supported(C, nfan, dtype):
        return False

    def mlp_chain_frag_supported(C, dtype):
        return False                                     # (the CPU stand-in keeps row-major weights)

    def feature_fusion_frag_supported(C, dtype):
        return False

    def pw_direct_supported(K, Cout, dtype):
        return False                                     # (fp16 only; the CPU stand-in is the fp32 wiring)

    def conv_narrow_supported(KH, KW, stride, Cin, Cout, dtype):
        return False                                     # (fp16 only)

    def mlp_chain(x, stages, res=None, res_stage=-1, carry=False, ln_eps=1e-5, ln_out=None, xcd_group_rows=0, fan=None, frag=False):
        t, ys = x.float(), []
        for s, (w, b, act, wsum) in enumerate(stages):
            a = F.layer_norm(t, (t.shape[-1],), eps=ln_eps) if wsum is not None else t
            if wsum is not None:
                assert torch.allclose(wsum, w.float().sum(1), atol=1e-4)
            y = ACTS[act](F.linear(a, w.float(), b))
            if s == res_stage:
                y = y + res.float()
            if carry and s == 2:
                y = y + ys[0]
            ys.append(y)
            t = y
        outs = [t.to(x.dtype)]
        if ln_out is not None:
            g, b, eps = ln_out
            outs.append(F.layer_norm(t.to(x.dtype).float(), (t.shape[-1],), g, b, eps).to(x.dtype))
        if fan is not None:
            fw, fb, fws = fan
            a = F.layer_norm(outs[0].float(), (t.shape[-1],), eps=ln_eps) if fws is not None else outs[0].float()
            outs.append(F.linear(a, fw.float(), fb).to(x.dtype))
        return outs[0] if len(outs) == 1 else tuple(outs)

    def feature_fusion_supported(C, dtype):
        return C in (128, 256)

    def conv_block_supported(C, H, W, dtype):
        return False                                     # (fp16 only; the engine needs the K5 v5 / K9 fragment packings for it)

    def row_attn_supported(C, heads, w, dtype):
        return C == 128 and heads in (1, 2) and 8 <= w <= 320       # (the kernel is fp16 only; the stand-in wires any dtype)

    def row_attn(x, heads, cross, weights, vectors, ln_eps=1e-5, ln_out_eps=None, xcd_hint=True):
        return row_attn_reference(x, heads, cross, weights, vectors, ln_eps, ln_out_eps)

    def feature_fusion(z0, z1, w1, b1, w2, bg, bf, z1_coarse=False):
        C = z0.shape[-1]
        if z1_coarse:
            z1 = F.interpolate(z1.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear", align_corners=False).permute(0, 2, 3, 1).to(z0.dtype)
        h = F.gelu(F.linear(torch.cat([z0.float(), z1.float()], -1), w1.float(), b1))
        g = torch.sigmoid(F.linear(h[..., :C], w2.float()[:, :C], bg)).clamp(0.01, 0.99)
        return (F.linear(h[..., C:], w2.float()[:, C:], bf) + g * z0.float() + (1 - g) * z1.float()).to(z0.dtype)

    def stem_mlp(x8, w0, b0, w1, b1):
        return F.linear(F.gelu(F.linear(x8.float(), w0, b0)), w1, b1).to(x8.dtype)

    def layernorm(x, out=None):
        return F.layer_norm(x.float(), (x.shape[-1],)).to(x.dtype)

    def groupnorm_nhwc(x, groups, gamma, beta, eps=1e-5):
        return F.group_norm(x.float().permute(0, 3, 1, 2), groups, gamma, beta, eps).permute(0, 2, 3, 1).contiguous().to(x.dtype)

    def resample2x(x, mode):
        xn = x.float().permute(0, 3, 1, 2)
        y = F.avg_pool2d(xn, 2) if mode == 0 else F.interpolate(xn, scale_factor=2, mode="bilinear", align_corners=False)
        return y.permute(0, 2, 3, 1).contiguous().to(x.dtype)

    def attention(q, k, v, heads, swap_halves=False, pe=None, scale=None):
        nb, N, C = q.shape
        d = C // heads
        sp = lambda t: t.float().reshape(t.shape[0], t.shape[1], heads, d).transpose(1, 2)
        qh, kh, vh = sp(q), sp(k), sp(v)
        if swap_halves:
            kh, vh = kh.roll(nb // 2, 0), vh.roll(nb // 2, 0)
        a = torch.softmax(qh @ kh.transpose(-1, -2) * (scale if scale is not None else d ** -0.5), -1)
        o = (a @ vh).transpose(1, 2).reshape(nb, N, C).to(q.dtype)
        if pe is None:
            return o
        px, py, gw, gh = pe
        ys, xs = torch.meshgrid(torch.arange(gh), torch.arange(gw), indexing="ij")
        xs, ys = xs.reshape(-1), ys.reshape(-1)
        dense = 0.5 * torch.cat([px[xs[:, None] - xs[None, :] + gw - 1], py[ys[:, None] - ys[None, :] + gh - 1]], 2)
        ps = torch.einsum("bhij,ijc->bhic", a, dense).transpose(1, 2).reshape(nb, N, heads * 32).to(q.dtype)
        return o, ps

    def convex_upsample(maps, logits, factor, scales=None, logit_up2=False, chan_out=None):
        lg = logits[..., :9].float().permute(0, 3, 1, 2)
        if logit_up2:
            lg = F.interpolate(lg, scale_factor=2, mode="bilinear", align_corners=False)
        wgt = lg.softmax(1)
        outs = []
        for m, s in zip(maps, scales or [1.0] * len(maps)):
            m = m.float().reshape(m.shape[0], 1, m.shape[-2], m.shape[-1])
            B, _, h, w = m.shape
            xp = F.pad(m, (1, 1, 1, 1), mode="replicate")
            n9 = torch.cat([xp[:, :, dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)], 1)
            if factor > 1:
                n9 = F.interpolate(n9, scale_factor=factor, mode="nearest")
            outs.append((n9 * wgt).sum(1, keepdim=True) * s)
        if chan_out is not None:
            chan_out.copy_(outs[0][:, 0].to(chan_out.dtype))
        return outs

    def image_prep(img0, img1, dtype):
        x = torch.cat([img0, img1], 0).float()
        x8 = torch.zeros(x.shape[0], x.shape[2], x.shape[3], 8, dtype=dtype)
        x8[..., 1:4] = ((x / 255.0 - 0.5) * 2).permute(0, 2, 3, 1).to(dtype)
        return x8

    def refine_prep(disp, conf, occ, mode, dtype):
        B, _, h, w = disp.shape
        small = torch.zeros(B, h, w, 8, dtype=dtype)
        if mode == 0:
            mask = (conf > 0.2).float()
            small[..., 0] = (disp / 1e2 * mask)[:, 0]
            small[..., 1] = torch.logit(mask * conf, eps=1e-1)[:, 0]
        else:
            small[..., 0] = (disp / 1e2)[:, 0]
            small[..., 1] = torch.logit(conf, eps=1e-2)[:, 0]
            small[..., 2] = torch.logit(occ, eps=1e-2)[:, 0]
        return small

    def global_update(upd, disp, conf, clamp0):
        mask = (conf > 0.2).float()
        d = mask * disp + (1 - mask) * (upd[..., 0].float().unsqueeze(1) * 1e2)
        return d.clamp(min=0) if clamp0 else d

    def refine_update(dco, disp, conf, occ, use_positivity, want_small=False):
        dt = dco.dtype
        dco = dco.float()
        d = disp + dco[..., 0].unsqueeze(1)
        c = torch.sigmoid(dco[..., 8].unsqueeze(1) + torch.logit(conf, eps=1e-2))
        o = torch.sigmoid(dco[..., 9].unsqueeze(1) + torch.logit(occ, eps=1e-2))
        if use_positivity:
            d = d.clamp(min=0)
        xs = torch.arange(d.shape[-1], dtype=torch.float32).reshape(1, 1, 1, -1)
        o = o * (xs - d >= 0)
        return (d, c, o, refine_prep(d, c, o, 1, dt)) if want_small else (d, c, o)

    def tanh(x):
        return torch.tanh(x)

    for f in (image_prep, refine_prep, global_update, refine_update, tanh, ln_corr, sinkhorn_regress, cv_lookup_into, conv2d, layernorm, groupnorm_nhwc, resample2x, attention, convex_upsample, mlp_chain, mlp_chain_supported, mlp_chain_ln_out_supported, mlp_chain_frag_supported, feature_fusion_frag_supported, pw_direct_supported, conv_narrow_supported, mlp_fan_supported, corr, cv_alloc, stem_mlp, feature_fusion, feature_fusion_supported, row_attn, row_attn_supported, conv_block_supported):
        setattr(ns, f.__name__, f)
    return ns
