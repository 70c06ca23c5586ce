#!/usr/bin/env python3
"""K2 (s2m2_sinkhorn_regress) alone at the BASELINE geometries, for profiler passes (tools/pmc_kernel.sh):   python tools/k2_only.py [c3] [c5]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip  # noqa: E402

CASES = {"c3": (256, 304, True), "c5": (512, 608, False), "c2": (120, 160, True), "c4": (256, 304, True), "c5pos": (512, 608, True), "c3nopos": (256, 304, False)}
names = [a for a in sys.argv[1:] if a in CASES] or ["c3", "c5", "c2", "c5pos", "c3nopos"]
for nm in names:
    h, w, pos = CASES[nm]
    g = torch.Generator(device="cuda").manual_seed(0)
    cv = hip.cv_alloc(1, h, w, torch.float16, "cuda")
    cv.copy_((torch.randn(1, h, w, w, device="cuda", generator=g) * 2.5 + 120).half())
    for _ in range(12):
        out = hip.sinkhorn_regress(cv, pos, 3)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        out = hip.sinkhorn_regress(cv, pos, 3)
    e1.record()
    torch.cuda.synchronize()
    print(f"{nm} h={h} w={w} positivity={pos}: {1e3 * e0.elapsed_time(e1) / reps:8.2f} us per launch (back to back, lib suffix '{os.environ.get('S2M2_LIB_SUFFIX', '')}')  "
          f"mean disp {float(out[0].float().mean()):.6f}", flush=True)
