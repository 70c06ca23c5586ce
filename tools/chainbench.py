#!/usr/bin/env python3
"""Time s2m2_mlp_chain (K9) against the separate K5 launches it replaces, hot (operands in L2) and cold (L2s evicted before
every call, as inside the pipeline).   python tools/chainbench.py"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.kbench import timeit_graph, timeit_graph_cold  # noqa: E402

SHAPES = [("1/32 x2 C256", 2 * 32 * 38, 256), ("1/16 x2 C256", 2 * 64 * 76, 256), ("1/8 x2 C128", 2 * 128 * 152, 128),
          ("1/4 x1 C128", 256 * 304, 128), ("1/4 x2 C128", 2 * 256 * 304, 128)]


def main():
    t8 = torch.zeros(64, device="cuda").half()
    print(f"launch floor: a 64-element s2m2_tanh inside the graph = {timeit_graph(lambda: hip.tanh(t8), 20, 3):.2f} us per dependent launch", flush=True)
    for name, rows, C in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        o = torch.randn(1, 1, rows, C, device="cuda", generator=g).half()
        z = torch.randn(1, 1, rows, C, device="cuda", generator=g).half()
        st = []
        for s in range(3):
            w = (torch.randn(C, C, 1, 1, device="cuda", generator=g) / math.sqrt(C)).half()
            wp = pack.pack_conv(w, torch.float16)
            st.append((wp, pack.pack_bias(torch.randn(C, device="cuda", generator=g), C), (0, 1, 0)[s], wp.float().sum(1).contiguous() if s == 1 else None))

        def chain3():
            return hip.mlp_chain(o, st, res=z, res_stage=0, carry=True)

        def sep3():
            z1 = hip.conv2d([o], st[0][0], st[0][1], 1, 1, C, epi=hip.EPI_ADD, aux0=z)
            h = hip.conv2d([z1], st[1][0], st[1][1], 1, 1, C, act=hip.ACT_GELU, ln_wsum=st[1][3])
            return hip.conv2d([h], st[2][0], st[2][1], 1, 1, C, epi=hip.EPI_ADD, aux0=z1)

        def chain2():
            return hip.mlp_chain(o, [(st[0][0], st[0][1], hip.ACT_RELU, None), (st[2][0], st[2][1], 0, None)])

        def sep2():
            u = hip.conv2d([o], st[0][0], st[0][1], 1, 1, C, act=hip.ACT_RELU)
            return hip.conv2d([u], st[2][0], st[2][1], 1, 1, C)

        fl3 = 3 * 2.0 * rows * C * C
        line = f"{name:14s}"
        for lbl, fn, fl in (("chain3", chain3, fl3), ("3xK5", sep3, fl3), ("chain2", chain2, fl3 * 2 / 3), ("2xK5", sep2, fl3 * 2 / 3)):
            th, tc = timeit_graph(fn, 20, 3), timeit_graph_cold(fn, 20, 3)
            line += f" | {lbl} hot {th:6.1f} us cold {tc:6.1f} us ({fl / tc / 1e6:5.1f} TF/s)"
        print(line, flush=True)


if __name__ == "__main__":
    main()
