#!/usr/bin/env python3
"""Per-block timeline of conv_frag_kernel (K5 v5) from shader-clock stamps (experiment build -DS2M2_FRAG_TRACE=1).
    S2M2_LIB_SUFFIX=_fragtrace S2M2_BUILD_DEFINES=-DS2M2_FRAG_TRACE=1 python -m s2m2_amd.build      (build container)
    S2M2_LIB_SUFFIX=_fragtrace python tools/frag_trace.py                                              (GPU box)"""
import ctypes
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402


def run(N, H, W, ci, co, kh, kw, act, epi, tile=0):
    lib = hip.load()
    lib.s2m2_debug_frag_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    x = torch.randn(N, H, W, ci, device="cuda").half()
    w = (torch.randn(co, ci, kh, kw, device="cuda") / math.sqrt(ci * kh * kw)).half()
    b = torch.randn(co, device="cuda")
    a0 = torch.rand(N, H, W, co, device="cuda").half() if epi else None
    wf, bp = pack.pack_conv_frag(w, torch.float16), pack.pack_bias(b, co)
    for _ in range(3):
        hip.conv2d([x], wf, bp, kh, kw, co, act=act, epi=epi, aux0=a0, korder=2, tile=tile)
    torch.cuda.synchronize()
    assert lib.s2m2_debug_frag_trace_clear() == 0                  # stamps of earlier launches (other geometries) out of the way
    hip.conv2d([x], wf, bp, kh, kw, co, act=act, epi=epi, aux0=a0, korder=2, tile=tile)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 4 * 8, dtype=np.uint64)
    assert lib.s2m2_debug_frag_trace(buf.ctypes.data, buf.nbytes) == 0
    t = buf.reshape(4096, 4, 8).astype(np.float64)
    nb = int((t[:, 0, 6] > 0).sum())                               # blocks that stamped their last slot in this launch's geometry
    t = t[:nb]
    us = 1e6 / 2.4e9

    def d(a, b_, q=50):
        return np.percentile((t[:, :, b_] - t[:, :, a]) * us, q)
    print(f"{N}x{H}x{W} {ci}->{co} k{kh}x{kw} act={act} epi={epi} tile={tile or 'auto'}: {nb} blocks x {co // 128} cout block(s) "
          f"(us @ 2.4 GHz ticks, per wave: p10 / median / p90; the shader clock is not synchronised across XCDs: differences inside one wave only)")
    for name, a, b_ in (("halo tile: loads + stash + barrier", 0, 1), ("K loop", 1, 2), ("barrier after the K loop", 2, 3),
                        ("aux requests + bias/activation + staging", 3, 4), ("barrier", 4, 5), ("aux combine + stores issued", 5, 6), ("whole block", 0, 6)):
        print(f"  {name:<44}{d(a, b_, 10):8.2f}{d(a, b_):8.2f}{d(a, b_, 90):8.2f}")


if __name__ == "__main__":
    run(1, 256, 304, 128, 128, 3, 3, 1, 0, tile=4)                 # round-2 geometry: 640 blocks of 4x32 pixels on 512 slots
    run(1, 256, 304, 128, 128, 3, 3, 1, 0)                         # the launcher's pick: 512 blocks of 4x40 pixels, one round
    run(1, 256, 304, 128, 128, 3, 3, 0, 1, tile=2)                 # residual add: 64-pixel blocks (round 2)
    run(1, 256, 304, 128, 128, 3, 3, 0, 1, tile=40)                # residual add: 160-pixel blocks, operand requested after the K loop
    run(1, 256, 304, 256, 128, 3, 3, 0, 0)
    run(2, 64, 76, 256, 256, 3, 3, 1, 0)
