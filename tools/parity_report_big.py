#!/usr/bin/env python3
"""Error tables of the HIP fp32 forward against the REFERENCE's own outputs at BASELINE configs[3] (L 1216x1024 refine_iter 3) and
configs[4] (XL 2432x2048, allow_negative) -- the comparison of tests/test_reference_golden_big.py, written out (GPU box).

    python tools/parity_report_big.py OUT.txt
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import parity_util as PU  # noqa: E402
import test_reference_golden_big as TB  # noqa: E402
from s2m2_amd.weights import seeded_state_dict, synthetic_pair  # noqa: E402


def stat(test, ref):
    d = (test.float() - ref.float()).abs().reshape(-1)
    n = d.numel()
    q = lambda f: float(d.kthvalue(max(1, int(round(f * n)))).values)
    return dict(n=n, scale=float(ref.float().abs().mean()), median=float(d.median()), p99=q(0.99), p999=q(0.999), max=float(d.max()),
                frac=float((d > TB.ATOL + TB.RTOL * ref.float().abs().reshape(-1)).float().mean()))


def run(name, out):
    g, c = TB._load(name)
    sd = seeded_state_dict(c["C"], 1, c["ntr"], c["seed"], gain=c["gain"])
    left, right = synthetic_pair(c["H"], c["W"], c["B"], c["disparity"], c["seed"])
    hout, hcap = PU.hip_forward(sd, c["C"], c["ntr"], c["ri"], left, right, False, use_positivity=c["pos"])
    rows = [("feature_tr_4x (every %d-th px)" % c["fsub"], stat(hcap["feature_tr_4x"][:, :, ::c["fsub"], ::c["fsub"]], TB._t(g["feature_tr_4x"]))),
            ("cv (every %d-th row)" % c["cvsub"], stat(hcap["cv"][:, ::c["cvsub"]], TB._t(g["cv"])))]
    am_ref = torch.as_tensor(g["argmax"].astype(np.int32))
    same = hcap["argmax"].int() == am_ref
    thr = max(1e-4, 4.0 * rows[1][1]["max"])
    sure = TB._sure(g, c, rows[1][1]["max"])
    for k in ("disp0", "conf0", "occ0"):
        rows.append((k + " (free running)", stat(hcap[k], TB._t(g[k]))))
    inj = {k: TB._t(g[k]) for k in ("disp0", "conf0", "occ0")}
    hout, hcap2 = PU.hip_forward(sd, c["C"], c["ntr"], c["ri"], left, right, False, inject=inj, use_positivity=c["pos"])
    rows.append(("disp_g (from the reference's DispInit)", stat(hcap2["disp_g"][..., ::c["gsub"], ::c["gsub"]], TB._t(g["disp_g"]))))
    for k, nm in enumerate(("disp", "occ", "conf")):
        rows.append((nm + " (final, every %d-th px)" % c["sub"], stat(hout[k][..., ::c["sub"], ::c["sub"]], TB._t(g[nm]))))
    out.write(f"{name}: C={c['C']} NTR={c['ntr']} {c['W']}x{c['H']} refine_iter={c['ri']} use_positivity={c['pos']} weights: seeded LeCun gain {c['gain']} "
              f"-- HIP fp32 vs the unmodified reference (tests/golden/{TB.FILES[name]})\n")
    out.write(f"tolerance per element: |err| <= {TB.ATOL:g} + {TB.RTOL:g}*|ref|\n")
    out.write(f"{'stage':<44}{'elements':>10}{'mean|ref|':>11}{'median err':>12}{'p99 err':>11}{'p99.9 err':>11}{'max err':>11}{'frac > tol':>12}\n")
    for nm, s in rows:
        out.write(f"{nm:<44}{s['n']:>10d}{s['scale']:>11.4g}{s['median']:>12.3e}{s['p99']:>11.3e}{s['p999']:>11.3e}{s['max']:>11.3e}{s['frac']:>12.3e}\n")
    out.write(f"integer argmax: agreement {float(same.float().mean()):.6f} on all pixels; 'sure' pixels (reference top-2 relative gap > {thr:.2e} = "
              f"max(1e-4, 4 x max cv error)): {float(sure.float().mean()):.5f} of all, mismatches there: {int((~same[sure]).sum())}\n\n")
    out.flush()


if __name__ == "__main__":
    with open(sys.argv[1], "w") as f:
        for name in TB.FILES:
            run(name, f)
    print(open(sys.argv[1]).read())
