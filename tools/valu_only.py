#!/usr/bin/env python3
"""Launch the K = 128 row-local kernels (K9 chain at 1/4 resolution x2, K10 at 1/4 resolution) and K2 at c3 a few times each, for a
rocprofv3 --pmc pass (tools/pmc_valu.sh): how busy are the VALU and the matrix pipe in the kernels DESIGN.md calls epilogue-bound?"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402


def main():
    g = torch.Generator(device="cuda").manual_seed(1)
    C, rows = 128, 2 * 256 * 304
    o = torch.randn(1, 1, rows, C, device="cuda", generator=g).half()
    z = torch.randn(1, 1, rows, C, device="cuda", generator=g).half()
    st = []
    for s in range(3):
        w = (torch.randn(C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).half()
        wp = pack.pack_conv(w, torch.float16)
        st.append((wp, pack.pack_bias(torch.randn(C, device="cuda", generator=g), C), (0, 1, 0)[s], wp.float().sum(1).contiguous() if s == 1 else None))
    for _ in range(6):
        hip.mlp_chain(o, st, res=z, res_stage=0, carry=True)
    # K10 at 1/4 resolution (one image)
    r1 = 256 * 304
    z0 = torch.randn(1, 1, r1, C, device="cuda", generator=g).half()
    z1 = torch.randn(1, 1, r1, C, device="cuda", generator=g).half()
    w1 = (torch.randn(3 * C, 2 * C, device="cuda", generator=g) / math.sqrt(2 * C)).half().contiguous()
    w2 = (torch.randn(C, 3 * C, device="cuda", generator=g) / math.sqrt(2 * C)).half().contiguous()
    b1, bg, bf = torch.zeros(3 * C, device="cuda"), torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
    for _ in range(6):
        hip.feature_fusion(z0, z1, w1, b1, w2, bg, bf)
    cv = (torch.randn(1, 256, 304, 304, device="cuda", generator=g) * 3 + 110).half()
    for _ in range(6):
        hip.sinkhorn_regress(cv, True, 3)
    torch.cuda.synchronize()
    print("valu_only: done")


if __name__ == "__main__":
    main()
