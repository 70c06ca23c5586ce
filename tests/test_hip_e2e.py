"""End-to-end GPU parity of ``s2m2_amd.S2M2.forward`` (fp32 mode) against golden outputs of the reference's PyTorch
CPU forward, plus batch independence and the fp16 deployment mode.

Tolerance (north_star): 1e-3 px on the fp32 disparity.  The reference itself moves by up to 1.3e-2 px on 0.2 % of the
pixels between 1 and 8 CPU threads when |disp| reaches hundreds of px with random weights (measured,
tests/test_oracle_golden.py), so the bound is |d| <= 1e-3 + 1e-4*|disp| on >= 99.5 % of pixels, max < 5e-2,
integer argmax bit exact wherever the reference's top-2 relative gap exceeds 1e-4."""
import pytest
import torch

import parity_util as PU
from conftest import T, load_golden
from s2m2_amd.model import S2M2
from s2m2_amd.weights import seeded_state_dict, synthetic_pair

pytestmark = pytest.mark.gpu


def _model(C, ntr, pos, ri, up, seed):
    m = S2M2(C, 1, ntr, use_positivity=bool(pos), output_upsample=bool(up), refine_iter=ri)
    m.load_state_dict(seeded_state_dict(C, 1, ntr, seed), strict=True)
    return m.cuda().eval()


@pytest.mark.parametrize("name", ["e2e_S_64x96_pos_r2", "e2e_S_96x160_neg_r1_b2", "e2e_S_64x64_pos_r1_up",
                                  "e2e_M_64x96_pos_r1", "e2e_L_64x96_pos_r2", "e2e_XL_64x64_pos_r1"])
def test_fp32_forward_vs_reference_golden(name):
    g = load_golden(name + ".npz")
    C, ntr, H, W, B, pos, ri, _, seed, up = [int(x) for x in g["cfg"]]
    m = _model(C, ntr, pos, ri, up, seed)
    cap = {}
    d, o, c = m(T(g["left"]).cuda(), T(g["right"]).cuda(), capture=cap)
    assert d.dtype == torch.float32 and tuple(d.shape) == g["disp"].shape
    d, o, c = d.cpu(), o.cpu(), c.cpu()
    assert float((cap["cv"].cpu() - T(g["cv"])).abs().max()) < 2e-3
    top1, top2 = g["top2"][..., 0], g["top2"][..., 1]
    sure = T((top1 - top2) > 1e-4 * top1)
    same = cap["argmax"].cpu() == T(g["argmax"])
    assert bool(same[sure].all())
    ref = T(g["disp"])
    err = (d - ref).abs()
    frac = float((err <= 1e-3 + 1e-4 * ref.abs()).float().mean())
    assert frac >= 0.995, (frac, float(err.max()))
    assert float(err.max()) < 5e-2
    assert float((o - T(g["occ"])).abs().max()) < 2e-4
    assert float((c - T(g["conf"])).abs().max()) < 2e-4
    # every stage boundary the golden file holds (reference hooks of tests/golden/make_golden.py), same 99.5 % criterion; 99.9 % when
    # no near-tie argmax differs (a flipped pixel moves disp0 by whole pixels and the refiners spread it)
    lim = 1e-3 if bool(same.all()) else 5e-3
    w4 = g["cv"].shape[-1]
    xs = torch.arange(w4, dtype=torch.float32).reshape(1, 1, 1, w4)
    clamp = (lambda t: t.clamp(min=0)) if pos else (lambda t: t)
    stages = {"disp0": T(g["disp0"]), "conf0": T(g["conf0"]), "occ0": T(g["occ0"]), "disp_g": clamp(T(g["disp_g_preclamp"]))}
    for k in ("feature_tr_4x", "feature_py_4x", "ctx", "hidden"):
        if k in g:
            stages[k] = T(g[k])
    for k in ("mask4x", "mask1x"):
        if k in g:
            cap[k] = cap[k][:, :, ::4, ::4]                                  # the golden keeps a strided sample of the logits
            stages[k] = T(g[k])
    for it in range(ri):
        dk = clamp(T(g[f"disp_it{it}_preclamp"]))
        stages[f"disp_it{it}"] = dk
        stages[f"conf_it{it}"] = T(g[f"conf_it{it}"])
        stages[f"occ_it{it}"] = T(g[f"occ_it{it}_premask"]) * (xs - dk >= 0)
        stages[f"corr1_it{it}"], stages[f"corr2_it{it}"] = T(g[f"corr1_it{it}"]), T(g[f"corr2_it{it}"])
    for k, ref_t in stages.items():
        st = PU.stats(cap[k], ref_t)
        # lookups: a 1e-4 px disparity difference times the slope of the cost curve (tens per px) -- 1 % allowed out of tolerance HERE,
        # on the free-running value; the operator itself is held to its own tolerance below
        assert st["finite"] and st["frac_out"] <= (1e-2 if k.startswith("corr") else lim), (k, st)
    # K3 at operator tolerance inside the end-to-end comparison: the lookups recomputed by the HIP kernel from the REFERENCE's own cost
    # volume and the reference's own disparity entering each iteration must reproduce the reference's corr tensors to one fp32 ulp of |cv|
    from s2m2_amd import hip as H
    cv_ref = T(g["cv"]).cuda()
    ulp2 = max(6e-5, 2.5 * 1.19e-7 * float(cv_ref.abs().max()))         # two fp32 ulps of |cv| (C = 384: |cv| ~ 380, ulp 3.05e-5)
    for it in range(ri):
        d_in = (stages["disp_g"] if it == 0 else stages[f"disp_it{it - 1}"]).float().cuda()
        c1, c2 = H.cv_lookup(cv_ref, d_in, 4)
        assert float((c1.cpu() - stages[f"corr1_it{it}"]).abs().max()) < ulp2, it
        assert float((c2.cpu() - stages[f"corr2_it{it}"]).abs().max()) < ulp2, it


def test_batch_independence_and_determinism():
    m = _model(128, 1, True, 1, False, 0)
    l, r = synthetic_pair(64, 96, 2, 8, 5)
    l, r = l.cuda(), r.cuda()
    d2, o2, c2 = m(l, r)
    d1a, _, _ = m(l[:1], r[:1])
    d1b, _, _ = m(l[1:], r[1:])
    assert float((d2 - torch.cat([d1a, d1b])).abs().max()) < 1e-3
    d2b, _, _ = m(l, r)                      # second forward: replays the refinement plans recorded by the first one (same launches)
    assert float((d2 - d2b).abs().max()) < 1e-3


def test_fp16_autocast_mode_tracks_fp32():
    """Deployment mode of the reference (autocast fp16, model_utils.py:76).  fp16 rounding of the cost volume legitimately
    flips near-tie matches (16 % under default init), so this is a sanity bound, not a parity claim:
    the median full-resolution disparity difference stays below 0.05 px and everything is finite."""
    m = _model(128, 1, True, 2, False, 0)
    l, r = synthetic_pair(128, 192, 1, 12, 9)
    l, r = l.cuda(), r.cuda()
    d32, o32, c32 = m(l, r)
    with torch.autocast("cuda", dtype=torch.float16):
        d16, o16, c16 = m(l, r)
    assert d16.dtype == torch.float32
    assert torch.isfinite(d16).all() and torch.isfinite(o16).all() and torch.isfinite(c16).all()
    assert float((d16 - d32).abs().median()) < 0.05
    assert float((c16 - c32).abs().median()) < 0.02


def test_smallest_legal_image():
    m = _model(128, 1, False, 1, False, 1)
    l, r = synthetic_pair(32, 32, 1, 2, 3)
    d, o, c = m(l.cuda(), r.cuda())
    assert tuple(d.shape) == (1, 1, 32, 32) and torch.isfinite(d).all()


def test_native_refine_step_is_bit_identical_and_rebinds_its_inputs(monkeypatch):
    """s2m2_refine_step (a recorded plan per refinement iteration, replayed from C++) against the Python-enqueued iteration: eager forwards on
    CHANGING images -- call 1 warms, call 2 records, calls 3+ replay with the externals (hidden, ctx, disp, conf, occ, cv) at new addresses and
    with new contents -- and the hipGraph path, which captures the plan's launches."""
    import torch
    from s2m2_amd import hip
    from s2m2_amd.model import S2M2
    from s2m2_amd.weights import seeded_state_dict, synthetic_pair
    sd = seeded_state_dict(128, 1, 1, 0)

    def build():
        m = S2M2(128, 1, 1, use_positivity=True, refine_iter=3)
        m.load_state_dict(sd, strict=True)
        return m.cuda().eval()

    pairs = [tuple(t.cuda() for t in synthetic_pair(96, 160, 1, 8 + 4 * k, k)) for k in range(4)]
    monkeypatch.setenv("S2M2_GRAPH", "0")
    monkeypatch.setenv("S2M2_REFINE_NATIVE", "0")
    ref_m = build()
    with torch.autocast("cuda", dtype=torch.float16):
        ref = [tuple(o.clone() for o in ref_m(l, r)) for l, r in pairs]
    monkeypatch.setenv("S2M2_REFINE_NATIVE", "1")
    nat_m = build()
    keep = []                                                      # hold earlier outputs: forces the allocator to hand out new addresses
    with torch.autocast("cuda", dtype=torch.float16):
        for k, (l, r) in enumerate(pairs):
            out = nat_m(l, r)
            keep.append(torch.empty(1 << 20, device="cuda"))
            for a, b in zip(out, ref[k]):
                assert torch.equal(a, b), f"forward {k}"
    eng = next(iter(nat_m._engines.values()))
    plans = [v for k_, v in {**eng._bufs, **eng._plans}.items() if isinstance(k_, tuple) and k_[0] == "refine_plan" and isinstance(v, tuple)]
    assert len(plans) == 3 and all(p[0].launches > 30 for p in plans), [p[0].launches for p in plans]
    monkeypatch.setenv("S2M2_GRAPH", "1")                          # graph path: two warm-ups, capture, replays
    g_m = build()
    with torch.autocast("cuda", dtype=torch.float16):
        for k, (l, r) in enumerate(pairs):
            out = g_m(l, r)
            for a, b in zip(out, ref[k]):
                assert torch.equal(a, b), f"graph forward {k}"


@pytest.mark.gpu
def test_coarse_level_stage_fusion_is_bit_identical(monkeypatch):
    """S2M2_COARSE_FUSE (engine.py): at 1/32, AvgPool + down_conv2 + the first attention block's Q | K | V as one K9 launch, and the last
    block's chain + the decoder's up_conv as one K9 launch -- the same arithmetic in the same order as the separate launches."""
    from s2m2_amd.weights import seeded_state_dict, synthetic_pair
    import parity_util as PU
    l, r = synthetic_pair(128, 192, 1, 16, 5)
    sd = seeded_state_dict(128, 1, 1, 0)
    res = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("S2M2_COARSE_FUSE", flag)
        res[flag] = PU.hip_forward(sd, 128, 1, 2, l, r, True)
    for a, b in zip(res["1"][0], res["0"][0]):
        assert torch.equal(a, b)
    assert torch.equal(res["1"][1]["feature_tr_4x"], res["0"][1]["feature_tr_4x"])
