"""End-to-end GPU parity at BASELINE.json's own configurations, every captured stage asserted against the pinned CPU oracle
executed inside the test (no /root/reference on the GPU box; the oracle is pinned to reference goldens by test_oracle_golden.py).

* configs[0]  S model 640x480 fp32 refine_iter=1, and the same with refine_iter=3          (free running)
* S model 1216x1024 fp32 refine_iter=1 (the benchmark geometry: ragged K1 tiles, 8-wave conv tiles, key-split attention, 64-row
  K9 / K10 tiles, which the 64x96 goldens never reach)
* configs[1]  S model 640x480 fp16 deployment mode, refine_iter=3, pinned to the oracle's emulation of the reference's autocast path

Criterion per stage: >= 99.9 % of the elements within |err| <= 1e-3 + 1e-4*|ref| (north_star: 1e-3 px fp32 disparity), integer
argmax bit exact wherever the oracle's top-2 relative gap exceeds 1e-4.  Measured values: profiles/r02/parity_c1_c3_c2.txt
(tools/parity_report.py prints the same statistics)."""
import pytest
import torch

import parity_util as PU
from oracle import s2m2_oracle as O
from s2m2_amd.weights import seeded_state_dict, synthetic_pair

pytestmark = pytest.mark.gpu

FRAC = 1e-3          # at most 0.1 % of the elements of a stage outside the tolerance


def _oracle(sd, left, right, ri, precision="fp32"):
    torch.set_num_threads(min(32, torch.get_num_threads()))
    cap = {}
    out = O.forward(sd, left, right, True, ri, False, cap, precision=precision)
    return out, cap


@pytest.mark.parametrize("refine_iter", [1, 3])
def test_fp32_640x480_every_stage(refine_iter):
    """BASELINE configs[0] (and refine_iter=3): measured 0 elements out of tolerance in every stage but the cost-volume lookups
    (3e-4 of them: a 7e-4 px disparity difference times the slope of the cost curve), final disparity max 5.4e-3 px."""
    sd = seeded_state_dict(128, 1, 1, 0)
    left, right = synthetic_pair(480, 640, 1, 32, 0)
    hout, hcap = PU.hip_forward(sd, 128, 1, refine_iter, left, right, False)
    oout, ocap = _oracle(sd, left, right, refine_iter)
    rows, am = PU.compare(hcap, hout, ocap, oout, refine_iter)
    assert len(rows) == len(PU.stage_list(refine_iter)), [r[0] for r in rows]
    assert am["mismatch_sure"] == 0 and am["agree_all"] >= 0.9995, am
    for name, _, s in rows:
        assert s["finite"], name
        assert s["frac_out"] <= FRAC, (name, s)
    st = PU.select(rows, ["disp", "occ", "conf", "cv", "feature_tr_4x"])
    assert st["cv"]["max"] < 1e-3 and st["feature_tr_4x"]["max"] < 2e-4
    if am["agree_all"] < 1.0:
        # a near-tie argmax (top-2 gap below 1e-4: not a 'sure' pixel) fell the other way: the maxima below are judged continuing
        # from the oracle's DispInit outputs, like the 1216x1024 test
        inj = {k: ocap[k] for k in ("disp0", "conf0", "occ0")}
        hout, hcap = PU.hip_forward(sd, 128, 1, refine_iter, left, right, False, inject=inj)
        rows, _ = PU.compare(hcap, hout, ocap, oout, refine_iter)
        st = PU.select(rows, ["disp", "occ", "conf"])
    assert st["disp"]["max"] < 2e-2 and st["disp"]["frac_out"] <= 1e-4, st["disp"]
    assert st["occ"]["max"] < 2e-4 and st["conf"]["max"] < 2e-4, st
    # the lookups at operator tolerance (one fp32 ulp of |cv| ~ 1e2 through the grid_sample coordinate round trip), from the oracle's
    # own disparities -- the free-running corr* stages above carry the upstream disparity error times the cost slope
    for name, err in PU.lookup_from_checker_disparity(ocap, refine_iter).items():
        assert err < 6e-5, (name, err)


@pytest.mark.parametrize("refine_iter,batch", [(1, 1), (3, 1), (1, 2)])
def test_fp32_1216x1024_every_stage(refine_iter, batch):
    """Benchmark geometry, also at the benchmark's refine_iter = 3 and with two pairs per launch sequence (B = 2: every (2B, ...) tensor
    and every grid doubles -- tile choices of K5 / K9 / K10 / K4 follow the row count).  Free running up to DispInit (the argmax may differ on the few pixels whose top-2 gap is below 1e-4: 3 of
    77 824 measured, each moving disp0 by tens of px, which a 2-D global attention then spreads); the stages after DispInit are
    judged continuing from the oracle's disp0 / conf0 / occ0 (teacher forcing through Engine's ``inject`` hook)."""
    ri = refine_iter
    sd = seeded_state_dict(128, 1, 1, 1)
    left, right = synthetic_pair(1024, 1216, batch, 48, 1)
    hout, hcap = PU.hip_forward(sd, 128, 1, ri, left, right, False)
    oout, ocap = _oracle(sd, left, right, ri)
    rows, am = PU.compare(hcap, hout, ocap, oout, ri)
    assert am["mismatch_sure"] == 0 and am["agree_all"] >= 0.999, am
    st = PU.select(rows, ["feature_py_4x", "feature_tr_4x", "cv", "ctx", "disp0", "conf0", "occ0"])
    for name in ("feature_py_4x", "feature_tr_4x", "cv", "ctx"):
        assert st[name]["frac_out"] == 0.0, (name, st[name])
    flipped = 1.0 - am["agree_all"]
    for name in ("disp0", "conf0", "occ0"):
        assert st[name]["frac_out"] <= flipped + 1e-5, (name, st[name])
    inj = {k: ocap[k] for k in ("disp0", "conf0", "occ0")}
    hout2, hcap2 = PU.hip_forward(sd, 128, 1, ri, left, right, False, inject=inj)
    rows2, _ = PU.compare(hcap2, hout2, ocap, oout, ri)
    assert len(rows2) == len(PU.stage_list(ri)), [r[0] for r in rows2]
    for name, _, s in rows2:
        if name in st:
            continue
        lim = 1e-2 if name.startswith("corr") else FRAC          # lookups: |d err| ~ 2e-4 px times a cost slope of ~100 / px
        assert s["finite"] and s["frac_out"] <= lim, (name, s)
    for name, err in PU.lookup_from_checker_disparity(ocap, ri).items():         # K3 at operator tolerance (|cv| ~ 1.3e2: one ulp = 1.5e-5)
        assert err < 6e-5, (name, err)
    fin = PU.select(rows2, ["disp", "occ", "conf"])
    # measured max 3.0e-2 px at |d| ~ 216 px (r = 1): the bound is ~2x that; the criterion proper is frac_out above
    assert fin["disp"]["max"] < 6e-2 and fin["conf"]["max"] < 1e-3, fin
    # occ carries a DISCONTINUITY: occ *= (x - disp >= 0) after every iteration (s2m2.py:179-180).  A 1/4-resolution pixel whose
    # x - disp is within the disparity tolerance of zero may fall on either side (measured at B = 2: two such pixels, |occ err| 0.65 on
    # their 12 x 12 full-resolution footprints, 1.1e-4 of the map); away from those knife edges the bound is 1e-3
    w4 = ocap["disp_g"].shape[-1]
    xs = torch.arange(w4, dtype=torch.float32).reshape(1, 1, 1, w4)
    edge = torch.zeros_like(ocap["disp_g"], dtype=torch.bool)
    for it in range(ri):
        edge |= (xs - ocap[f"disp_it{it}"]).abs() < 2e-3
    assert float(edge.float().mean()) < 1e-2              # (mostly column 0, where the positivity clamp makes x - disp exactly 0 on both sides)
    grown = torch.nn.functional.max_pool2d(edge.float(), 3, 1, 1)                       # the convex upsampling mixes 3 x 3 neighbours
    keep = torch.nn.functional.interpolate(grown, scale_factor=4, mode="nearest") == 0
    keep = keep & (torch.nn.functional.max_pool2d((~keep).float(), 3, 1, 1) == 0)          # ... and the 1x sharpening 3 x 3 once more
    occ_err = (hout2[1] - oout[1]).abs()
    assert float(occ_err[keep].max()) < 1e-3, (float(occ_err[keep].max()), fin["occ"])


def test_fp16_640x480_pinned_to_autocast_emulation():
    """BASELINE configs[1].  The fp16 mode is pinned to the oracle's emulation of the reference's CUDA-autocast numerics:

    1. free running, the transformer features stay as close to the emulation as the emulation is to fp32 (rounding noise of
       ~60 fp16 layers; no discrete decisions up to there);
    2. K1 from the SAME fp16 features is bit exact up to single fp16 ulps of the accumulated sums, and K2's integer argmax agrees
       on >= 99.8 % of the pixels (the reference rounds the probabilities to fp16 before its argmax and accumulates the 5-tap
       window in fp16 -- K2 keeps fp32, so disp0 differs by the reference's own fp16 quantisation, <= 0.25 px);
    3. everything after DispInit, continued from the SAME cv / disp0 / conf0 / occ0, ends within 0.01 px (median) / 0.3 px (p99)
       of the emulation's full-resolution disparity.
    Measured: profiles/r02/parity_c1_c3_c2.txt."""
    ri = 3
    sd = seeded_state_dict(128, 1, 1, 0)
    left, right = synthetic_pair(480, 640, 1, 32, 0)
    o16, c16 = _oracle(sd, left, right, ri, "fp16")
    o32, c32 = _oracle(sd, left, right, ri, "fp32")
    # 1. free running
    hout, hcap = PU.hip_forward(sd, 128, 1, ri, left, right, True)
    assert all(torch.isfinite(t).all() for t in hout)
    rows, am = PU.compare(hcap, hout, c16, o16, ri)
    ref_rows, ref_am = PU.compare(c16, o16, c32, o32, ri)
    mine, theirs = PU.select(rows, ["feature_py_4x", "feature_tr_4x", "ctx"]), PU.select(ref_rows, ["feature_py_4x", "feature_tr_4x", "ctx"])
    for name in mine:
        assert mine[name]["p999"] <= 1.5 * theirs[name]["p999"], (name, mine[name], theirs[name])
    assert am["agree_all"] >= ref_am["agree_all"] - 0.02, (am, ref_am)          # argmax flips: not more than fp16 itself causes
    # 2. K1 + K2 from the emulation's features
    h2, hc2 = PU.hip_forward(sd, 128, 1, ri, left, right, True, inject={"feature_tr_4x": c16["feature_tr_4x"]})
    r2, am2 = PU.compare(hc2, h2, c16, o16, ri)
    s2 = PU.select(r2, ["cv", "disp0", "conf0", "occ0"])
    d = (hc2["cv"].float() - c16["cv"].float()).abs()
    assert float(d.max()) <= 0.125 and float((d > 0).float().mean()) <= 2e-3, (float(d.max()), float((d > 0).float().mean()))
    assert am2["agree_all"] >= 0.998 and am2["mismatch_sure"] <= 10, am2
    assert s2["disp0"]["p99"] <= 0.3 and s2["conf0"]["p999"] <= 1e-2 and s2["occ0"]["p999"] <= 5e-3, s2
    # 3. stages after DispInit from the emulation's DispInit outputs
    inj = {k: c16[k] for k in ("cv", "disp0", "conf0", "occ0")}
    h3, hc3 = PU.hip_forward(sd, 128, 1, ri, left, right, True, inject=inj)
    r3, _ = PU.compare(hc3, h3, c16, o16, ri)
    s3 = PU.select(r3, ["disp", "occ", "conf", f"disp_it{ri - 1}"])
    assert s3["disp"]["median"] <= 1e-2 and s3["disp"]["p99"] <= 0.3, s3["disp"]
    assert s3[f"disp_it{ri - 1}"]["p99"] <= 0.05, s3
    assert s3["conf"]["p999"] <= 5e-3 and s3["occ"]["p999"] <= 5e-3, s3


_sharp_tokens = PU.sharp_tokens


def test_fp16_640x480_sharp_matches_free_running():
    """The fp16 deployment mode held to a TIGHT bound end to end: with sharp, unambiguous matches injected as feature_tr_4x on both
    sides there are no near-tie argmax flips -- the only thing that separates two valid fp16 forwards of this randomly initialised
    model -- so the free-running HIP fp16 forward must reproduce the autocast emulation's integer argmax exactly, disp0 to the
    reference's own fp16 quantisation, and the final maps to fp16 rounding noise through the refiners (no teacher forcing after the
    injection point)."""
    ri = 3
    sd = seeded_state_dict(128, 1, 1, 0)
    left, right = synthetic_pair(480, 640, 1, 32, 0)
    tok = _sharp_tokens(128, 120, 160, (12, 5, 23), 4)
    inj = {"feature_tr_4x": tok}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    c16, c32 = {}, {}
    o16 = O.forward(sd, left, right, False, ri, False, c16, precision="fp16", inject=inj)
    o32 = O.forward(sd, left, right, False, ri, False, c32, precision="fp32", inject=inj)
    top = c32["prob"].topk(2, 3).values
    assert float(((top[..., 0] - top[..., 1]) / top[..., 0]).min()) > 0.5            # the construction: every argmax is unambiguous
    assert bool((c16["argmax"] == c32["argmax"]).all())
    hout, hcap = PU.hip_forward(sd, 128, 1, ri, left, right, True, inject=inj, use_positivity=False)
    assert all(torch.isfinite(t).all() for t in hout)
    assert bool((hcap["argmax"].long() == c16["argmax"].long()).all()), "integer argmax must be bit exact when no near tie exists"
    rows, am = PU.compare(hcap, hout, c16, o16, ri)
    ref_rows, _ = PU.compare(c16, o16, c32, o32, ri)
    st = PU.select(rows, ["cv", "disp0", "conf0", "occ0", "disp_g", f"disp_it{ri - 1}", "disp", "occ", "conf"])
    ref = PU.select(ref_rows, ["disp", "occ", "conf", f"disp_it{ri - 1}"])
    print({k: (v["median"], v["p99"], v["max"]) for k, v in st.items()})
    print("emulation vs fp32:", {k: (v["median"], v["p99"], v["max"]) for k, v in ref.items()})
    assert st["cv"]["max"] <= 0.125                                                   # one fp16 ulp of |cv| <= 128 ... 256
    # K2 keeps fp32 where the reference's autocast rounds probabilities and the 5-tap sums to fp16: |j| <= 160 -> ulp 0.125
    assert st["disp0"]["max"] <= 0.16 and st["conf0"]["max"] <= 2e-3 and st["occ0"]["max"] <= 2e-3, st
    # free running from there: no discrete decision is left, what remains is fp16 rounding noise amplified by the randomly initialised
    # refiners (the emulation itself sits median 0.065 / p99 0.64 px from the fp32 forward of the same features: measured with the
    # oracle in both modes) -- the HIP forward must be no further from the emulation than the emulation is from fp32, at the median
    # and at p99, for every final map (with argmax flips out of the picture this yardstick is 20x tighter than
    # tests/test_fp16_reference_autocast.py's: p99 0.64 px instead of 13.5 px)
    rows32, _ = PU.compare(hcap, hout, c32, o32, ri)
    st32 = PU.select(rows32, ["disp", "occ", "conf", f"disp_it{ri - 1}"])
    print("HIP fp16 vs fp32:", {k: (v["median"], v["p99"], v["max"]) for k, v in st32.items()})
    # ... and MUCH closer to the fp32 forward than the reference's own fp16 mode is (measured: disparity median 8.9e-4 / p99 0.059 px
    # against 0.066 / 0.64 px for the autocast emulation: the kernels keep fp32 where autocast rounds to fp16 between ops): held to
    # 2x the measured values, an order of magnitude inside the reference's own fp16-vs-fp32 distance
    assert st32["disp"]["median"] <= 2e-3 and st32["disp"]["p99"] <= 0.12, st32["disp"]
    assert st32[f"disp_it{ri - 1}"]["p99"] <= 3e-3 and st32["occ"]["p99"] <= 4e-4 and st32["conf"]["p99"] <= 4e-4, st32
    # (measured: HIP vs emulation 0.0658 / 0.645 px against emulation vs fp32 0.0655 / 0.638 px; occ 2.4e-4 / 1.6e-3 vs 2.3e-4 / 1.6e-3;
    # conf 7.8e-4 / 4.0e-3 vs 7.8e-4 / 4.0e-3 -- two independent fp16 roundings of the same fp32 forward; the bound is 1.15x)
    for name in ("disp", "occ", "conf", f"disp_it{ri - 1}"):
        assert st[name]["median"] <= 1.15 * ref[name]["median"] + 1e-4 and st[name]["p99"] <= 1.15 * ref[name]["p99"] + 1e-3, (name, st[name], ref[name])
