"""Host-logic test (no GPU): the engine's stage wiring, weight plumbing and state_dict naming, with the HIP
entry points replaced by CPU stand-ins (tests/fake_hip.py: plain PyTorch CPU ops / the oracle).  This checks the Python around the kernels only --
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


from fake_hip import make as _fake_hip


@pytest.mark.parametrize("name", ["e2e_S_64x96_pos_r2", "e2e_S_64x64_pos_r1_up", "e2e_M_64x96_pos_r1"])
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
