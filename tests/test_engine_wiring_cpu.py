"""Host-logic test (no GPU): the engine's stage wiring, weight plumbing and state_dict naming, with the three HIP
entry points replaced by CPU stand-ins built from the oracle.  This checks the Python around the kernels only --
the kernels themselves are covered by the -m gpu tests through the real C ABI."""
import types

import numpy as np
import pytest
import torch

from conftest import T, load_golden
from oracle import s2m2_oracle as O
import s2m2_amd.engine as engine_mod
from s2m2_amd.model import S2M2, build_model
from s2m2_amd.weights import seeded_state_dict


def _fake_hip():
    def ln_corr(tokens, g, b, cv_dtype=None):
        return O.ln_corr(tokens.permute(0, 3, 1, 2).float(), g, b)

    def sinkhorn_regress(cv, pos, ot_iter=3, want_argmax=False):
        d, c, o, ind = O.regress(O.sinkhorn_prob(cv.float(), pos, ot_iter))
        return (d, c, o, ind.int()) if want_argmax else (d, c, o)

    def cv_lookup(cv, disp, radius=4, channels_last=False, out_dtype=torch.float32):
        c1, c2 = O.cv_lookup(cv.float(), disp.float(), radius)
        if channels_last:
            c1, c2 = c1.permute(0, 2, 3, 1), c2.permute(0, 2, 3, 1)
        return c1.to(out_dtype), c2.to(out_dtype)
    return types.SimpleNamespace(load=lambda: None, ln_corr=ln_corr, sinkhorn_regress=sinkhorn_regress, cv_lookup=cv_lookup)


@pytest.mark.parametrize("name", ["e2e_S_64x96_pos_r2", "e2e_S_64x64_pos_r1_up"])
def test_engine_wiring_matches_reference_golden(monkeypatch, name):
    monkeypatch.setattr(engine_mod, "hip", _fake_hip())
    g = load_golden(name + ".npz")
    C, ntr, H, W, B, pos, ri, _, seed, up = [int(x) for x in g["cfg"]]
    m = S2M2(C, 1, ntr, use_positivity=bool(pos), output_upsample=bool(up), refine_iter=ri)
    m.load_state_dict(seeded_state_dict(C, 1, ntr, seed), strict=True)
    eng = engine_mod.Engine(m, torch.float32)
    cap = {}
    d, o, c = eng.run(T(g["left"]), T(g["right"]), cap)
    assert d.shape == g["disp"].shape
    assert float((cap["cv"] - T(g["cv"])).abs().max()) < 5e-4
    assert bool((cap["argmax"] == T(g["argmax"])).all())
    assert float((d - T(g["disp"])).abs().max()) < 2e-3
    assert float((o - T(g["occ"])).abs().max()) < 1e-4
    assert float((c - T(g["conf"])).abs().max()) < 1e-4


def test_model_is_drop_in_for_state_dict_and_rejects_cpu():
    m = build_model("S", use_positivity=True, refine_iter=1)
    sd = seeded_state_dict(128, 1, 1, 3)
    m.my_load_state_dict({"state_dict": sd}["state_dict"])
    assert list(m.state_dict().keys()) == list(sd.keys())
    assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
    # shape-mismatch tolerant like the reference's my_load_state_dict
    bad = dict(sd)
    bad["ctx_feat.0.weight"] = torch.zeros(3, 3)
    m.my_load_state_dict(bad)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 3, 32, 32), torch.zeros(1, 3, 32, 32))
