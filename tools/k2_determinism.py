#!/usr/bin/env python3
"""K2 run-to-run determinism: the same cost volume through s2m2_sinkhorn_regress N times, every output compared bit for bit with the
first run (plus the reference golden once).  python tools/k2_determinism.py [--runs 200]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip  # noqa: E402


def check(name, cv, pos, runs):
    ref = None
    bad = 0
    for r in range(runs):
        out = hip.sinkhorn_regress(cv, pos, 3, want_argmax=True)
        torch.cuda.synchronize()
        out = [t.clone() for t in out]
        if ref is None:
            ref = out
            continue
        if not all(torch.equal(a, b) for a, b in zip(ref, out)):
            bad += 1
            if bad <= 3:
                for k, (a, b) in enumerate(zip(ref, out)):
                    d = (a.float() - b.float()).abs()
                    print(f"    run {r} output {k}: {int((d > 0).sum())} elements differ, max {float(d.max()):.3e}")
    print(f"{name}: {bad} of {runs - 1} repeat runs differ from the first", flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--runs", type=int, default=200)
    a = ap.parse_args()
    hip.load()
    total = 0
    gdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    for name in ("op_dispinit_pos", "op_dispinit_neg"):
        g = np.load(os.path.join(gdir, name + ".npz"))
        cv = torch.from_numpy(g["cv"]).cuda()
        total += check(f"{name} {tuple(cv.shape)}", cv, bool(g["cfg"][4]), a.runs)
    gen = torch.Generator(device="cuda").manual_seed(0)
    for shape, dt in (((1, 120, 160, 160), torch.float16), ((1, 64, 304, 304), torch.float16), ((1, 16, 608, 608), torch.float32)):
        cv = (torch.randn(*shape, device="cuda", generator=gen) * 8).to(dt)
        for pos in (True, False):
            total += check(f"random {shape} {dt} pos={pos}", cv, pos, max(20, a.runs // 4))
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
