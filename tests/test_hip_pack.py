"""s2m2_pack_frag (csrc/pack.hip: the fragment orders of the direct-form kernels produced by the library) against the torch formulas of
s2m2_amd/pack.py, bit for bit, on the layer shapes of the four model sizes -- the library path is what the engine runs on a GPU, the torch path is
what the CPU tests pin against the layouts' definitions (tests/test_pack_cpu.py)."""
import math

import pytest
import torch

from s2m2_amd import pack

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from s2m2_amd import hip as h
    h.load()
    return h


def _both(fn, monkeypatch):
    a = fn()
    monkeypatch.setenv("S2M2_PACK_TORCH", "1")
    b = fn()
    monkeypatch.delenv("S2M2_PACK_TORCH")
    return a, b


@pytest.mark.parametrize("rows,C", [(128, 128), (384, 128), (768, 256), (192, 192), (1152, 384), (64, 32)])
def test_chain_frag_native_equals_torch(hip, monkeypatch, rows, C):
    w = torch.randn(rows, C, device="cuda").half()
    a, b = _both(lambda: pack.chain_frag(w), monkeypatch)
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("cout,k", [(128, 384), (16, 48), (96, 32), (64, 256), (36, 128), (8, 8), (256, 264)])
def test_pw_frag_native_equals_torch(hip, monkeypatch, cout, k):
    w = torch.randn(cout, k, device="cuda").half()
    a, b = _both(lambda: pack.pw_frag(w), monkeypatch)
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("cout,cin,ntap", [(160, 8, 9), (64, 16, 25), (96, 96, 9), (8, 128, 9), (32, 256, 9), (48, 48, 9), (16, 144, 9)])
def test_narrow_frag_native_equals_torch(hip, monkeypatch, cout, cin, ntap):
    if cin >= 128 and cin % 64:
        pytest.skip("not a K12 shape")
    w = torch.randn(cout, ntap * cin, device="cuda").half()
    a, b = _both(lambda: pack.narrow_frag(w, ntap), monkeypatch)
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("r,k", [(16, 48), (9, 48), (32, 96), (8, 40)])
def test_head_frag_native_equals_torch(hip, monkeypatch, r, k):
    w = torch.randn(r, k, device="cuda").half()
    a, b = _both(lambda: pack.head_frag(w), monkeypatch)
    assert a.shape == b.shape and torch.equal(a, b)


@pytest.mark.parametrize("C", [128, 192, 256, 384])
def test_fusion_frag_native_equals_torch(hip, monkeypatch, C):
    w1 = torch.randn(3 * C, 2 * C, device="cuda").half()
    w2 = torch.randn(C, 3 * C, device="cuda").half()
    a, b = _both(lambda: pack.fusion_frag(w1, w2), monkeypatch)
    assert a.numel() == 9 * C * C and torch.equal(a, b)


@pytest.mark.parametrize("co,ci,kh,kw,splits", [(128, 128, 3, 3, None), (256, 256, 3, 1, [(128, 128), (128, 128)]), (128, 192, 3, 3, None),
                                                 (384, 256, 3, 3, [(128, 128), (128, 128)]), (128, 320, 1, 3, None), (256, 384, 3, 3, None),
                                                 (192, 192, 3, 3, None), (192, 384, 1, 3, [(192, 192), (192, 192)]), (576, 192, 3, 3, [(96, 96), (96, 96)])])
def test_conv_frag_native_equals_torch(hip, monkeypatch, co, ci, kh, kw, splits):
    w = (torch.randn(co, ci, kh, kw, device="cuda") / math.sqrt(ci * kh * kw))
    a, b = _both(lambda: pack.pack_conv_frag(w, torch.float16, splits), monkeypatch)
    assert a.shape == b.shape and torch.equal(a, b)


def test_pack_frag_rejects_bad_descriptors(hip):
    w = torch.randn(40, 48, device="cuda").half()
    with pytest.raises(RuntimeError, match="at most 32"):
        hip.pack_frag(hip.PACK_HEAD, w)
    with pytest.raises(RuntimeError, match="taps"):
        hip.pack_frag(hip.PACK_CONV_FRAG, torch.randn(32, 100, device="cuda").half(), ntap=9)
    with pytest.raises(ValueError):
        hip.pack_frag(hip.PACK_ROWS, w.float())
