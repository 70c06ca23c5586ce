#!/usr/bin/env python3
"""A batch of B pairs through the drop-in module as ONE batched launch sequence (S2M2_PAIR_STREAMS=0) or as two chunks of ceil(B / 2) pairs on
two side streams (S2M2_PAIR_STREAMS=2, the default): ms per pair.   python tools/batch_modes.py   (runs both modes in subprocesses)"""
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from s2m2_amd.model import build_model
    from s2m2_amd.weights import noise_pair
    m = build_model("S", use_positivity=True, refine_iter=3).cuda().eval()
    for H, W, B in ((1024, 1216, 2), (1024, 1216, 4), (480, 640, 2), (480, 640, 4), (480, 640, 8)):
        l, r = (t.cuda() for t in noise_pair(H, W, B, seed=5))
        with torch.autocast("cuda", dtype=torch.float16):
            for _ in range(4):
                out = m(l, r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 10
            for _ in range(n):
                out = m(l, r)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        print(f"S2M2_PAIR_STREAMS={os.environ.get('S2M2_PAIR_STREAMS', '2')}  {W}x{H} B={B}: {1e3 * dt / B:7.3f} ms per pair  finite={bool(torch.isfinite(out[0]).all())}", flush=True)
else:
    for mode in ("0", "2", "0", "2"):
        subprocess.run([sys.executable, __file__, "--child"], env=dict(os.environ, S2M2_PAIR_STREAMS=mode))
