#!/usr/bin/env python3
"""Does the forward's period depend on how far the host runs ahead of the GPU?  ms per forward of the S model's graph replay with a host
synchronisation after every 1 / 4 / 20 / 200 forwards, same process, same box.     python tools/runahead_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd.model import build_model  # noqa: E402
from s2m2_amd.weights import noise_pair  # noqa: E402

m = build_model("S", use_positivity=True, refine_iter=3).cuda().eval()
left, right = (t.cuda() for t in noise_pair(1024, 1216, 1, 0))


def fwd():
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        return m(left, right)


for _ in range(3):
    fwd()
torch.cuda.synchronize()
for rep in range(2):
    for every in (1, 4, 20, 200):
        total = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.time()
        e0.record()
        for i in range(total):
            fwd()
            if (i + 1) % every == 0:
                torch.cuda.synchronize()
        e1.record()
        torch.cuda.synchronize()
        host = time.time() - t0
        print(f"rep {rep}  sync after every {every:3d} forwards: {e0.elapsed_time(e1) / total:8.3f} ms per forward (events)   {1e3 * host / total:8.3f} ms (host clock)")
