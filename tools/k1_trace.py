#!/usr/bin/env python3
"""Timeline of K1 (ln_corr) inside the CUs: per-phase shader-clock stamps of every wave (experiment build with
-DS2M2_LNCORR_TRACE=1 -> libs2m2_hip_k1trace.so; see the K1_T() stamps in csrc/ln_corr.hip).

    S2M2_LIB_SUFFIX=_k1trace S2M2_BUILD_DEFINES=-DS2M2_LNCORR_TRACE=1 python -m s2m2_amd.build      (build container)
    S2M2_LIB_SUFFIX=_k1trace python tools/k1_trace.py [case]                                          (GPU box)
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip  # noqa: E402

CASES = {"c2": (128, 120, 160), "c3": (128, 256, 304), "c4": (256, 256, 304), "c5": (384, 512, 608)}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    case = args[0] if args else "c3"
    prenorm = "--prenorm" in sys.argv           # s2m2_corr: tokens already normalised (no LayerNorm phases)
    aligned = "--aligned" in sys.argv           # cost-volume rows on 128-byte lines (hip.cv_alloc)
    C, h, w = CASES[case]
    lib = hip.load()
    lib.s2m2_debug_k1_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    torch.manual_seed(0)
    feat = (torch.randn(2, h, w, C, device="cuda") * 1.5).half()
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    out = hip.cv_alloc(1, h, w, torch.float16, "cuda", aligned=True) if aligned else torch.empty((1, h, w, w), device="cuda", dtype=torch.float16)
    if prenorm:
        feat = torch.nn.functional.layer_norm(feat.float(), (C,)).half()

    def launch():
        if prenorm:
            hip.corr(feat, out=out)
        elif aligned:
            raise SystemExit("--aligned needs --prenorm (s2m2_ln_corr writes dense volumes)")
        else:
            hip.ln_corr(feat, g, b, out=out)
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    buf = np.zeros(1024 * 16 * 16, dtype=np.uint64)
    assert lib.s2m2_debug_k1_trace(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(1024, 16, 16)[:h].astype(np.float64)
    nw = int((t[0, :, 0] > 0).sum())
    t = t[:, :nw]
    t0 = t[:, :, 0].min()
    span = t[:, :, 15].max() - t0
    # s_memtime ticks are shader cycles on gfx950 (MI355X_MICROARCH.md); if the span looks like a 100 MHz counter, say so
    hz = 100e6 if span < 5e3 else 2.4e9
    scale = 1e6 / hz
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        launch()
    ev1.record()
    torch.cuda.synchronize()
    print(f"variant: {'s2m2_corr (normalised tokens)' if prenorm else 's2m2_ln_corr'}, cv pitch {out.stride(2)}")
    print(f"K1 timeline {case}: {h} blocks x {nw} waves; raw span {span:.0f} ticks, read as {hz / 1e6:.0f} MHz; eager back-to-back "
          f"{ev0.elapsed_time(ev1) * 1e3 / 20:.1f} us per launch (instrumented build); microseconds since the first wave's entry")
    # (the shader clock is not synchronised across XCDs: only differences inside one wave are meaningful)
    # per-block view: duration of each phase for the median wave
    def d(a, b_, q=50):
        return np.percentile((t[:, :, b_] - t[:, :, a]) * scale, q)
    print(f"{'per-wave duration (us @ 2.4 GHz)':<36}{'p10':>8}{'median':>8}{'p90':>8}{'max':>8}")
    for name, a, b_ in (("prologue (affine -> LDS, barrier)", 0, 1), ("issue of all token loads", 1, 2), ("wait + LayerNorm left tokens", 2, 3),
                        ("LayerNorm right tokens", 3, 4), ("block barrier wait", 4, 5), ("MFMA + staging + store issue", 5, 15),
                        ("entry -> last store issued", 0, 15)):
        print(f"{name:<36}{d(a, b_, 10):8.2f}{d(a, b_):8.2f}{d(a, b_, 90):8.2f}{d(a, b_, 100):8.2f}")
    nbytes = 2 * h * w * C * 2 + h * w * w * 2
    print(f"bytes per launch {nbytes / 1e6:.1f} MB: loads {2 * h * w * C * 2 / 1e6:.1f} MB in issue+wait = {d(1, 3):.2f} us -> "
          f"{2 * h * w * C * 2 / d(1, 3) / 1e6:.1f} TB/s;  stores {h * w * w * 2 / 1e6:.1f} MB issued in {d(5, 15):.2f} us -> "
          f"{h * w * w * 2 / d(5, 15) / 1e6:.1f} TB/s (issue rate; the queue drains after the last stamp)")


if __name__ == "__main__":
    main()
