"""The benchmark geometry against the REFERENCE itself: tests/golden/e2e_S_1216x1024_fp32_r1_sub4.npz holds the unmodified reference's
DispInit outputs and its final maps (every 4th pixel) for the S model at 1216x1024, fp32, refine_iter 1 (tests/golden/make_golden_c3.py).

* CPU: the oracle reproduces them (the fp32 pin of tests/test_oracle_golden.py, extended to the size the benchmark runs);
* GPU: the HIP fp32 forward -- DispInit free running, the refinement stages continued from the reference's own disp0 / conf0 / occ0."""
import os

import numpy as np
import pytest
import torch

from oracle import s2m2_oracle as O
from s2m2_amd.weights import seeded_state_dict, synthetic_pair

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2e_S_1216x1024_fp32_r1_sub4.npz")


def _setup():
    g = np.load(GOLD)
    C, ntr, H, W, B, pos, ri, disparity, seed = [int(v) for v in g["cfg"]]
    return g, seeded_state_dict(C, 1, ntr, seed), C, ntr, ri, bool(pos), synthetic_pair(H, W, B, disparity, seed), int(g["sub"])


def _t(a):
    return torch.as_tensor(np.asarray(a)).float()


def test_oracle_reproduces_the_reference_at_the_benchmark_geometry():
    g, sd, C, ntr, ri, pos, (left, right), sub = _setup()
    torch.set_num_threads(min(8, torch.get_num_threads()))
    cap = {}
    out = O.forward(sd, left, right, pos, ri, False, cap, precision="fp32")
    for k in ("disp0", "conf0", "occ0"):
        d = (cap[k].float() - _t(g[k])).abs()
        assert float((d > 1e-3).float().mean()) <= 1e-4, (k, float(d.max()))       # (a near-tie argmax may fall the other way)
    for k, name in enumerate(("disp", "occ", "conf")):
        d = (out[k][..., ::sub, ::sub].float() - _t(g[name])).abs()
        lim = 1e-3 + 1e-4 * _t(g[name]).abs()
        # (disparities average 216 px here: the tolerance is 1e-3 + 1e-4 |d| ~ 2e-2 px, the median error a few fp32 ulps of |d|)
        assert float((d > lim).float().mean()) <= 1e-3 and float(d.median()) < (2e-3 if name == "disp" else 1e-4), \
            (name, float(d.max()), float((d > lim).float().mean()), float(d.median()))


@pytest.mark.gpu
def test_hip_fp32_against_the_reference_at_the_benchmark_geometry():
    import parity_util as PU
    g, sd, C, ntr, ri, pos, (left, right), sub = _setup()
    hout, hcap = PU.hip_forward(sd, C, ntr, ri, left, right, False)
    flipped = 0.0
    for k in ("disp0", "conf0", "occ0"):                                            # DispInit, free running
        d = (hcap[k].float() - _t(g[k])).abs()
        frac = float((d > 1e-3 + 1e-4 * _t(g[k]).abs()).float().mean())
        flipped = max(flipped, frac)
        assert frac <= 2e-4, (k, frac, float(d.max()))
    inj = {k: _t(g[k]) for k in ("disp0", "conf0", "occ0")}                         # refinement from the reference's own DispInit outputs
    hout, _ = PU.hip_forward(sd, C, ntr, ri, left, right, False, inject=inj)
    for k, name in enumerate(("disp", "occ", "conf")):
        ref = _t(g[name])
        d = (hout[k][..., ::sub, ::sub] - ref).abs()
        frac = float((d > 1e-3 + 1e-4 * ref.abs()).float().mean())
        assert frac <= 1e-3 and float(d.median()) < (2e-3 if name == "disp" else 1e-4), (name, frac, float(d.median()), float(d.max()))
        assert float(d.max()) < (0.1 if name == "disp" else 1e-3), (name, float(d.max()))
