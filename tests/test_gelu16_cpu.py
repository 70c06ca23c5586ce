"""The GELU of fp16 destinations (csrc/epilogue.h: fast_gelu16x2) restated in numpy float32 from the coefficients IN THE HEADER, against
x * Phi(x) in float64: the error bounds the header states hold, and rounded to fp16 it differs from the correctly rounded GELU less often
than the A&S 7.1.26 form it replaced."""
import os
import re

import numpy as np
from scipy.special import ndtr

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "s2m2_amd", "csrc", "epilogue.h")


def _coefficients():
    src = open(HDR).read()
    body = src[src.index("float2_t fast_gelu16x2(float2_t x)"):]
    body = body[:body.index("__builtin_amdgcn_exp2f")]
    c = [float(v) for v in re.findall(r"\(float2_t\)\((-?[0-9.]+(?:e-?[0-9]+)?)f\)", body)]
    assert len(c) == 8, c                                            # c7, c6 of the first fma, then c5 .. c0
    return c


def _gelu16(x, c):
    a = np.abs(x).astype(np.float32)
    p = np.float32(c[0]) * a + np.float32(c[1])
    for ck in c[2:]:
        p = (p * a + np.float32(ck)).astype(np.float32)            # (fma on the device: one rounding fewer per step)
    return (np.maximum(x, np.float32(0)) - a * np.exp2(p).astype(np.float32)).astype(np.float32)


def _gelu_as(x):                                                     # the A&S form (fast_gelu), as in the header
    ax = np.abs(x).astype(np.float32)
    t = (np.float32(1) / (np.float32(0.3275911 * 0.70710678118654752440) * ax + np.float32(1))).astype(np.float32)
    pl = np.float32(0.5 * 1.061405429) * t + np.float32(0.5 * -1.453152027)
    for ck in (0.5 * 1.421413741, 0.5 * -0.284496736, 0.5 * 0.254829592):
        pl = (pl * t + np.float32(ck)).astype(np.float32)
    q = pl * t * np.exp2(ax * ax * np.float32(-0.5 * 1.44269504088896340736)).astype(np.float32)
    return (np.maximum(x, np.float32(0)) - ax * q).astype(np.float32)


def test_fast_gelu16_error_bounds_and_fp16_faithfulness():
    c = _coefficients()
    assert c[0] < 0                                                  # leading coefficient: P keeps falling past the fitted interval
    x = np.linspace(-8, 8, 2_000_001).astype(np.float32)
    ref = x.astype(np.float64) * ndtr(x.astype(np.float64))
    new, old = _gelu16(x, c), _gelu_as(x)
    assert np.abs(new - ref).max() <= 1.0e-6                         # header: absolute <= 8.1e-7 over all x
    neg = x < 0
    rel = np.abs(new[neg] - ref[neg]) / np.maximum(np.abs(ref[neg]), 6e-8)
    assert rel[x[neg] >= -6.5].max() <= 1.0e-5                       # header: relative 4.7e-6 of the a * Phi(-a) term on the fitted interval
    h_ref = ref.astype(np.float16)
    flips_new, flips_old = np.mean(new.astype(np.float16) != h_ref), np.mean(old.astype(np.float16) != h_ref)
    assert flips_new <= 2.5e-3 and flips_new < flips_old, (flips_new, flips_old)
    ulp = np.abs(new.astype(np.float16).view(np.int16).astype(np.int32) - h_ref.view(np.int16).astype(np.int32))
    assert ulp.max() <= 1
    big = np.array([10.0, 50.0, 1000.0, 65504.0, -10.0, -50.0, -1000.0, -65504.0], np.float32)
    with np.errstate(over="ignore", under="ignore"):
        out = _gelu16(big, c)
    # far outside the fit: x on the positive side, a vanishing negative number (-10 Phi(-10) = -7.6e-23) or zero on the other, never NaN
    assert np.array_equal(out[:4], big[:4]) and np.all(np.abs(out[4:]) < 1e-20) and np.all(out[5:] == 0)
