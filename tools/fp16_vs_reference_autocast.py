#!/usr/bin/env python3
"""HIP fp16 forward (and the oracle's fp16 mode) against the reference's own fp16 run (CPU autocast golden, tests/golden/make_golden_fp16.py):
median / p90 / p99 of the final maps next to the reference's fp16-vs-fp32 spread.     python tools/fp16_vs_reference_autocast.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import test_fp16_reference_autocast as T  # noqa: E402
from oracle import s2m2_oracle as O  # noqa: E402


def main():
    g, sd, C, ntr, ri, pos, left, right = T._setup()
    rows = [("reference fp16 (CPU autocast) vs reference fp32", [T._stats(g[n + "_fp16"], g[n + "_fp32"]) for n in ("disp", "occ", "conf")])]
    o16 = O.forward(sd, left, right, pos, ri, False, {}, precision="fp16")
    rows.append(("oracle fp16 mode vs reference fp16", [T._stats(o16[k], g[n + "_fp16"]) for k, n in enumerate(("disp", "occ", "conf"))]))
    if torch.cuda.is_available():
        import parity_util as PU
        h, _ = PU.hip_forward(sd, C, ntr, ri, left, right, True)
        rows.append(("HIP fp16 vs reference fp16", [T._stats(h[k], g[n + "_fp16"]) for k, n in enumerate(("disp", "occ", "conf"))]))
        rows.append(("HIP fp16 vs reference fp32", [T._stats(h[k], g[n + "_fp32"]) for k, n in enumerate(("disp", "occ", "conf"))]))
    print("S model 640x480 refine_iter 3 (BASELINE configs[1]); median / p90 / p99 of |difference|: disparity [px] | occlusion | confidence")
    for name, st in rows:
        print(f"{name:50s} " + " | ".join(f"{m:.4f} {p90:.4f} {p99:.4f}" for m, p90, p99 in st))


if __name__ == "__main__":
    main()
