"""Generate golden vectors by running the UNMODIFIED reference (PyTorch CPU, fp32) in the build container.

    python tests/golden/make_golden.py            # needs /root/reference; writes tests/golden/*.npz

The reference ships no tests, golden vectors or weights, so parity is pinned on
outputs of the reference itself: seeded weights (s2m2_amd.weights.seeded_state_dict -- independent of
torch RNG) + seeded synthetic inputs, stage boundaries captured with hooks.  The .npz files travel to
the GPU box; /root/reference does not.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import s2m2.core.model.s2m2 as ref_s2m2  # noqa: E402  (reference, read-only)
import s2m2.core.model.submodules as ref_sub  # noqa: E402
import s2m2.core.model.attentions as ref_attn  # noqa: E402
import s2m2.core.model.refinenet as ref_refine  # noqa: E402
import s2m2.core.model.utils as ref_utils  # noqa: E402

from s2m2_amd.weights import seeded_state_dict, synthetic_pair, draw_tensor  # noqa: E402

torch.set_num_threads(8)
META = dict(torch=torch.__version__, threads=torch.get_num_threads())


def _np(d):
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in d.items()}


def _save(name, d):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **_np(d))
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB  keys={sorted(d)}")


def e2e(name, C, ntr, H, W, B, use_pos, refine_iter, disparity, seed, output_upsample=False, keep_features=True):
    sd = seeded_state_dict(C, 1, ntr, seed)
    model = ref_s2m2.S2M2(C, 1, ntr, use_positivity=use_pos, output_upsample=output_upsample, refine_iter=refine_iter).eval()
    model.load_state_dict(sd, strict=True)
    left, right = synthetic_pair(H, W, B, disparity, seed)
    cap = {}

    # --- hooks (no reference source is modified; wrappers only observe) ---
    model.transformer.register_forward_hook(lambda m, i, o: cap.__setitem__("feature_tr_4x", o))
    model.feat_pyramid.register_forward_hook(lambda m, i, o: cap.__setitem__("feature_py_4x", o[0]))
    model.disp_init.register_forward_hook(
        lambda m, i, o: cap.update(disp0=o[0], conf0=o[1], occ0=o[2], cv=o[3]))
    ot = model.disp_init._optimal_transport

    def ot_wrap(attn):
        p = ot(attn)
        cap["prob_unmasked"] = p
        return p
    model.disp_init._optimal_transport = ot_wrap
    model.global_refiner.register_forward_hook(lambda m, i, o: cap.__setitem__("disp_g_preclamp", o))
    model.ctx_feat.register_forward_hook(lambda m, i, o: cap.__setitem__("ctx", o))
    it = [0]

    def refiner_hook(m, i, o):
        k = it[0]
        cap[f"disp_it{k}_preclamp"], cap[f"conf_it{k}"], cap[f"occ_it{k}_premask"] = o[1], o[2], o[3]
        cap["hidden"] = o[0]
        it[0] += 1
    model.refiner.register_forward_hook(refiner_hook)
    model.upsample_mask_4x_refine.register_forward_hook(lambda m, i, o: cap.__setitem__("mask4x", o))
    model.upsample_mask_1x.register_forward_hook(lambda m, i, o: cap.__setitem__("mask1x", o))

    class CVSpy(ref_sub.CostVolume):
        n = 0

        def __call__(self, disp):
            c1, c2 = super().__call__(disp)
            cap[f"corr1_it{CVSpy.n}"], cap[f"corr2_it{CVSpy.n}"] = c1, c2
            CVSpy.n += 1
            return c1, c2
    old = ref_s2m2.CostVolume
    ref_s2m2.CostVolume = CVSpy
    try:
        with torch.no_grad():
            d, o, c = model(left, right)
    finally:
        ref_s2m2.CostVolume = old

    w = cap["cv"].shape[-1]
    P = cap.pop("prob_unmasked")
    if use_pos:
        P = P.masked_fill(torch.ones(w, w, dtype=torch.bool).triu(1), 0)
    top2 = P.topk(2, dim=3).values
    out = dict(disp=d, occ=o, conf=c, argmax=P.argmax(dim=3).to(torch.int32), top2=top2, left=left, right=right,
               cfg=np.array([C, ntr, H, W, B, int(use_pos), refine_iter, disparity, seed, int(output_upsample)]))
    for k, v in cap.items():
        if k in ("feature_tr_4x", "feature_py_4x", "hidden", "ctx", "mask4x", "mask1x") and not keep_features:
            continue
        out[k] = v
    for k in ("mask4x", "mask1x"):          # large (B,9,H,W): keep a strided sample only
        if k in out:
            out[k] = out[k][:, :, ::4, ::4].contiguous()
    _save(name, out)


def op_dispinit(name, C, h, w, B, use_pos, seed, noise=0.25, shift_max=12):
    """DispInit on synthetic, sharply matchable features: right = left shifted by a per-row disparity."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(B, C, h, w + shift_max, generator=g)
    f0 = base[..., shift_max:]
    f1 = torch.empty(B, C, h, w)
    for y in range(h):
        d = int(torch.randint(0, shift_max + 1, (1,), generator=g))
        f1[:, :, y] = base[:, :, y, shift_max - d: shift_max - d + w]      # f1[x - d] == f0[x]
    f1 = f1 + noise * torch.randn(B, C, h, w, generator=g)
    # a few exactly duplicated right columns -> exercises "first max wins" on near ties
    f1[:, :, 0, 5] = f1[:, :, 0, 4]
    feat = torch.cat([f0, f1], 0) * 1.7 + 0.3
    m = ref_sub.DispInit(C, 3, use_pos).eval()
    m.layer_norm.weight.data = draw_tensor("disp_init.layer_norm.weight", (C,), seed)
    m.layer_norm.bias.data = draw_tensor("disp_init.layer_norm.bias", (C,), seed)
    cap = {}
    ot = m._optimal_transport
    m._optimal_transport = lambda a: cap.setdefault("P", ot(a))
    with torch.no_grad():
        disp, conf, occ, cv = m(feat)
    P = cap["P"]
    if use_pos:
        P = P.masked_fill(torch.ones(w, w, dtype=torch.bool).triu(1), 0)
    _save(name, dict(feat=feat, gamma=m.layer_norm.weight.data, beta=m.layer_norm.bias.data, disp=disp, conf=conf,
                     occ=occ, cv=cv, argmax=P.argmax(3).to(torch.int32), top2=P.topk(2, 3).values,
                     cfg=np.array([C, h, w, B, int(use_pos), seed])))


def op_lookup(name, h, w, B, seed):
    g = torch.Generator().manual_seed(seed)
    cv = torch.randn(B, h, w, w, generator=g) * 20 + 100
    xs = torch.arange(w, dtype=torch.float32).reshape(1, 1, 1, w)
    disp = torch.rand(B, 1, h, w, generator=g) * (w + 12) - 6          # includes d<0 and x-d<0 (out of range)
    disp[:, :, 0, :8] = torch.tensor([0.0, 1.0, 2.0, 0.5, 1.5, 3.25, -1.0, 7.0])   # integral / half cases
    coords = torch.arange(w, dtype=torch.float32).reshape(1, 1, w, 1).repeat(B, h, 1, 1)
    fn = ref_sub.CostVolume(cv, coords, radius=4)
    c1, c2 = fn(disp)
    _save(name, dict(cv=cv, disp=disp, corr1=c1, corr2=c2))


def op_attention(name, dim, heads, B, hh, ww, seed):
    """SelfAttn with and without PE, CrossAttn -- on (B, N=hh*ww, dim) tokens."""
    g = torch.Generator().manual_seed(seed)
    N = hh * ww
    x = torch.randn(B, N, dim, generator=g)
    y = torch.randn(B, N, dim, generator=g)
    out = dict(x=x, y=y, cfg=np.array([dim, heads, B, hh, ww]))
    for use_pe in (False, True):
        m = ref_attn.SelfAttn(dim, heads, 1, use_pe).eval()
        pref = "sa_pe" if use_pe else "sa"
        for k, v in m.state_dict().items():
            t = draw_tensor(f"{name}.{pref}.{k}", tuple(v.shape), seed, bias_std=0.1)
            v.copy_(t)
            out[f"{pref}.{k}"] = t
        pe = ref_utils.get_pe(hh, ww, 32, torch.float32, "cpu") if use_pe else None
        with torch.no_grad():
            out[f"{pref}.out"] = m(x, pe)
    m = ref_attn.CrossAttn(dim, heads, 1).eval()
    for k, v in m.state_dict().items():
        t = draw_tensor(f"{name}.ca.{k}", tuple(v.shape), seed, bias_std=0.1)
        v.copy_(t)
        out[f"ca.{k}"] = t
    with torch.no_grad():
        ox, oy = m(x, y)
    out["ca.out_x"], out["ca.out_y"] = ox, oy
    out["pe"] = ref_utils.get_pe(hh, ww, 32, torch.float32, "cpu")
    _save(name, out)


def op_gru_upsample(name, C, h, w, B, seed):
    g = torch.Generator().manual_seed(seed)
    out = {}
    m = ref_refine.ConvGRU(C, C, 3).eval()
    for k, v in m.state_dict().items():
        t = draw_tensor(f"{name}.gru.{k}", tuple(v.shape), seed)
        v.copy_(t)
        out[f"gru.{k}"] = t
    hid = torch.tanh(torch.randn(B, C, h, w, generator=g))
    x = torch.randn(B, C, h, w, generator=g)
    with torch.no_grad():
        out["gru.out"] = m(hid, x)
    out["gru.h"], out["gru.x"] = hid, x
    # convex upsampling: S2M2.upsample4x / upsample1x are methods; instantiate a tiny model shell
    shell = ref_s2m2.S2M2.__new__(ref_s2m2.S2M2)
    disp = torch.rand(B, 1, h, w, generator=g) * 40
    m4 = torch.randn(B, 9, 4 * h, 4 * w, generator=g) * 3
    out["up.disp"], out["up.mask4"] = disp, m4
    out["up.out4"] = ref_s2m2.S2M2.upsample4x(shell, disp, m4)
    full = torch.rand(B, 1, 4 * h, 4 * w, generator=g) * 160
    m1 = torch.randn(B, 9, 4 * h, 4 * w, generator=g) * 3
    out["up.full"], out["up.mask1"] = full, m1
    shell.output_upsample = False
    out["up.out1"] = ref_s2m2.S2M2.upsample1x(shell, full, m1)
    shell.output_upsample = True
    out["up.out1_2x"] = ref_s2m2.S2M2.upsample1x(shell, full, m1)
    _save(name, out)


if __name__ == "__main__":
    only = sys.argv[1] if len(sys.argv) > 1 else ""                # python make_golden.py [substring]: regenerate matching files only
    _all = e2e, op_dispinit, op_lookup, op_attention, op_gru_upsample
    if only:
        def _filtered(fn):
            return lambda name, *a, **k: fn(name, *a, **k) if only in name else None
        e2e, op_dispinit, op_lookup, op_attention, op_gru_upsample = [_filtered(f) for f in _all]
    e2e("e2e_S_64x96_pos_r2.npz", 128, 1, 64, 96, 1, True, 2, 8, 0)
    e2e("e2e_S_96x160_neg_r1_b2.npz", 128, 1, 96, 160, 2, False, 1, 12, 1, keep_features=False)
    e2e("e2e_S_64x64_pos_r1_up.npz", 128, 1, 64, 64, 1, True, 1, 4, 2, output_upsample=True, keep_features=False)
    # the other published model sizes (README of the reference: M = 192 x 2 transformers, L = 256 x 3, XL = 384 x 3)
    e2e("e2e_M_64x96_pos_r1.npz", 192, 2, 64, 96, 1, True, 1, 8, 3, keep_features=False)
    e2e("e2e_L_64x96_pos_r2.npz", 256, 3, 64, 96, 1, True, 2, 6, 4, keep_features=False)
    e2e("e2e_XL_64x64_pos_r1.npz", 384, 3, 64, 64, 1, True, 1, 5, 5, keep_features=False)
    op_dispinit("op_dispinit_pos.npz", 128, 6, 72, 2, True, 3)
    op_dispinit("op_dispinit_neg.npz", 64, 5, 40, 1, False, 4)
    op_lookup("op_lookup.npz", 5, 40, 2, 5)
    op_attention("op_attention.npz", 64, 8, 2, 5, 7, 6)
    op_gru_upsample("op_gru_upsample.npz", 32, 6, 10, 2, 7)
    print(META)
