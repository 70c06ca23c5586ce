"""Robustness check, part of the default GPU run (S2M2_TEST_LDS_POISON=0 skips it): every kernel family is launched right after
s2m2_debug_poison_lds has left quiet-NaN patterns in the LDS of every CU, and must still produce the outputs of an unpoisoned run bit for
bit -- i.e. no kernel reads an LDS word it never wrote (the K2 padding bug of round 2 was such a read).  Three geometries so that the
kernels that keep state in LDS across passes are all reached: S at 192x256 (v3 / v5 64-pixel tiles, K9 / K10 at C = 128 / 256, K2 with 16
lanes per row), S at 256x1600 (w = 400: K2 with 32 lanes per row, key-split attention) and L at 128x192 (C = 256 / 512 tiles, d = 64)."""
import os

import pytest
import torch

from s2m2_amd.weights import seeded_state_dict, synthetic_pair

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("S2M2_TEST_LDS_POISON") == "0", reason="disabled: S2M2_TEST_LDS_POISON=0")]


@pytest.fixture(scope="module")
def hip():
    from s2m2_amd import hip as h
    h.load()
    return h


@pytest.mark.parametrize("C,ntr,H,W,pos", [(128, 1, 192, 256, True), (128, 1, 256, 1600, False), (256, 3, 128, 192, True)])
def test_forward_is_unchanged_when_every_kernel_starts_on_poisoned_lds(hip, monkeypatch, C, ntr, H, W, pos):
    from s2m2_amd.model import S2M2
    sd = seeded_state_dict(C, 1, ntr, 0, gain=0.9 if C > 128 else 1.0)
    left, right = synthetic_pair(H, W, 1, 16, 0)
    m = S2M2(C, 1, ntr, use_positivity=pos, refine_iter=2)
    m.load_state_dict(sd, strict=True)
    m = m.cuda().eval()
    monkeypatch.setenv("S2M2_GRAPH", "0")
    for fp16 in (False, True):
        with torch.autocast("cuda", dtype=torch.float16, enabled=fp16):
            ref = [t.clone() for t in m(left.cuda(), right.cuda())]
            # eager launches with a poison launch in front of every C-ABI call
            orig = {}
            lib = hip.load()

            def wrap(name):
                f = getattr(lib, name)

                def g(*a):
                    hip.poison_lds()
                    return f(*a)
                return g
            names = [n for n in hip.SIGNATURES if n.startswith("s2m2_") and n not in ("s2m2_debug_poison_lds", "s2m2_last_error", "s2m2_version")
                     and "supported" not in n and "workspace" not in n and "event" not in n and "kernel_name" not in n]
            for n in names:
                orig[n] = getattr(lib, n)
                setattr(lib, n, wrap(n))
            try:
                out = m(left.cuda(), right.cuda())
                torch.cuda.synchronize()
            finally:
                for n, f in orig.items():
                    setattr(lib, n, f)
        for a, b in zip(ref, out):
            assert torch.isfinite(b).all() and torch.equal(a, b)
