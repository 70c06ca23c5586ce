#!/usr/bin/env python3
"""K9 at SHORT row counts: the row-major form (weights staged through LDS, a block barrier per 64-byte K chunk) against the direct form
(s2m2_chain_desc.weight_frag: fragments from global memory into the MFMA operand registers), with and without the next attention's
Q | K | V projection as fan-out stages, against chain + separate K5 projection.   python tools/chainbench_direct.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.kbench import timeit_graph, timeit_graph_cold  # noqa: E402

SHAPES = [("1/32 x1 C256", 32 * 38, 256), ("1/32 x2 C256", 2 * 32 * 38, 256), ("1/16 x1 C256", 64 * 76, 256), ("1/16 x2 C256", 2 * 64 * 76, 256),
          ("1/8 x1 C128", 128 * 152, 128), ("1/8 x2 C128", 2 * 128 * 152, 128)]


def main():
    for name, rows, C in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        o = torch.randn(1, 1, rows, C, device="cuda", generator=g).half()
        z = torch.randn(1, 1, rows, C, device="cuda", generator=g).half()
        st = []
        for s in range(3):
            w = (torch.randn(C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).half()
            wp = pack.pack_conv(w, torch.float16)
            st.append((wp, pack.pack_bias(torch.randn(C, device="cuda", generator=g), C), (0, 1, 0)[s], wp.float().sum(1).contiguous() if s == 1 else None))
        fst = [(pack.chain_frag(w), b, a, ws) for w, b, a, ws in st]
        wq = pack.pack_conv((torch.randn(3 * C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).half(), torch.float16)
        bq = pack.pack_bias(torch.randn(3 * C, device="cuda", generator=g), 3 * C)
        wsq = wq.float().sum(1).contiguous()
        wqf = pack.chain_frag(wq)

        def chain3():
            return hip.mlp_chain(o, st, res=z, res_stage=0, carry=True)

        def direct3():
            return hip.mlp_chain(o, fst, res=z, res_stage=0, carry=True, frag=True)

        def chain3_qkv():
            y = hip.mlp_chain(o, st, res=z, res_stage=0, carry=True)
            return hip.conv2d([y], wq, bq, 1, 1, 3 * C, ln_wsum=wsq)

        def chain3_fan():
            return hip.mlp_chain(o, st, res=z, res_stage=0, carry=True, fan=(wq, bq, wsq))

        def direct3_fan():
            return hip.mlp_chain(o, fst, res=z, res_stage=0, carry=True, fan=(wqf, bq, wsq), frag=True)

        a, b = chain3_fan(), direct3_fan()
        assert all(torch.equal(u, v) for u, v in zip(a, b))
        line = f"{name:14s}"
        for lbl, fn in (("chain3", chain3), ("direct3", direct3), ("chain3 + K5 qkv", chain3_qkv), ("chain3+fan", chain3_fan), ("direct3+fan", direct3_fan)):
            th, tc = timeit_graph(fn, 20, 3), timeit_graph_cold(fn, 20, 3)
            line += f" | {lbl} hot {th:5.1f} cold {tc:5.1f}"
        print(line + "  (us)", flush=True)


if __name__ == "__main__":
    main()
