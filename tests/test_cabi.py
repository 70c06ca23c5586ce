"""CPU-side checks of the drop-in boundary: libs2m2_hip.so loads and exports every symbol of include/s2m2_hip.h.
No compute calls (there is no GPU in the build container); argument validation paths only."""
import os
import re

import pytest

from s2m2_amd import hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from s2m2_amd.build import build
    build(verbose=False)
    return hip.load()


def _declared():
    txt = open(os.path.join(ROOT, "include", "s2m2_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return set(re.findall(r"\b(s2m2_[a-z0-9_]+)\s*\(", txt))


def test_every_declared_symbol_is_exported_and_bound(lib):
    names = _declared()
    assert len(names) >= 7
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/s2m2_hip.h but not exported"
    assert names == set(hip.SIGNATURES)


def test_version_and_error_reporting(lib):
    assert lib.s2m2_version() >= 100
    assert lib.s2m2_ln_corr(None, None, None, None, 1, 1, 8, 128, 1, 1, None) != 0
    assert b"null pointer" in lib.s2m2_last_error()
    assert lib.s2m2_sinkhorn_regress(None, None, None, None, None, 1, 1, 8, 3, 1, 1, None, None) != 0
    assert lib.s2m2_cv_lookup(None, None, None, None, 1, 1, 8, 4, 1, 0, 0, 0, 0, None) != 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", "/nonexistent/libs2m2_hip.so")
    with pytest.raises(RuntimeError, match="no PyTorch fallback"):
        hip.load()
