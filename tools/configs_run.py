#!/usr/bin/env python3
"""Run every BASELINE.json configuration that fits one GPU through the drop-in module (fp16 autocast) and time it:
c2 S 640x480, c3 S 1216x1024, c4 L 1216x1024, c5 XL 2432x2048 allow_negative (+ M as a bonus).  Checks finiteness, prints ms/pair."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd.model import build_model
from s2m2_amd.weights import noise_pair
CFG = [("c2", "S", 480, 640, True), ("c3", "S", 1024, 1216, True), ("c4", "L", 1024, 1216, True), ("M", "M", 1024, 1216, True),
       ("c5", "XL", 2048, 2432, False)]
only = sys.argv[1:]
for name, mt, H, W, pos in CFG:
    if only and name not in only:
        continue
    try:
        m = build_model(mt, use_positivity=pos, refine_iter=3).cuda().eval()
        l, r = noise_pair(H, W, 1, 0)
        l, r = l.cuda(), r.cuda()
        with torch.autocast("cuda", dtype=torch.float16):
            for _ in range(3):
                d, o, c = m(l, r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 5
            for _ in range(n):
                d, o, c = m(l, r)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        ok = bool(torch.isfinite(d).all() and torch.isfinite(o).all() and torch.isfinite(c).all())
        print(f"{name}: {mt}-model {W}x{H} fp16 refine_iter=3 positivity={pos}: {dt * 1e3:8.2f} ms/pair  {1 / dt:7.2f} pairs/s  finite={ok} "
              f"disp[min,max]=[{float(d.min()):.1f},{float(d.max()):.1f}]  peak mem {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB", flush=True)
        del m, l, r, d, o, c
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        print(f"{name}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
