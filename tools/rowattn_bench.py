#!/usr/bin/env python3
"""K13 (s2m2_row_attn: one 1-D attention step per launch) against the launch triple it replaces -- K9 fan-out Q | K | V, K4, K9 chain -- on the S
model's 1/4 and 1/8 level shapes (fp16), hipGraph-replayed like the forward.     python tools/rowattn_bench.py [c2]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.kbench import timeit_graph  # noqa: E402

C = 128
SHAPES = [("L0 1216x1024", 2, 256, 304, 1), ("L1 1216x1024", 2, 128, 152, 2), ("L0 two pairs", 4, 256, 304, 1), ("L1 2432x2048", 2, 256, 304, 2)]
if "c2" in sys.argv:
    SHAPES = [("L0 640x480", 2, 120, 160, 1), ("L1 640x480", 2, 60, 80, 2)]
g = torch.Generator(device="cuda").manual_seed(0)
qkv = (torch.randn(3 * C, C, device="cuda", generator=g) / math.sqrt(C)).half().contiguous()
bqkv = torch.zeros(3 * C, device="cuda")
rest = [(torch.randn(C, C, device="cuda", generator=g) / math.sqrt(C)).half().contiguous() for _ in range(3)]
b0, b2 = torch.randn(C, device="cuda") * 0.3, torch.randn(C, device="cuda") * 0.3
ws = qkv.float().sum(1).contiguous()
weights = pack.rowattn_pack(torch.cat([qkv] + rest, 0))
vectors = pack.rowattn_vectors((ws[:C], ws[C:2 * C], ws[2 * C:], rest[1].float().sum(1)), (None, None, None, None, b0, b2))
fq = pack.chain_frag(qkv)
st = [(pack.chain_frag(rest[0]), None, hip.ACT_NONE, None), (pack.chain_frag(rest[1]), b0, hip.ACT_GELU, rest[1].float().sum(1).contiguous()), (pack.chain_frag(rest[2]), b2, hip.ACT_NONE, None)]
for name, nimg, h, w, heads in SHAPES:
    x = torch.randn(nimg, h, w, C, device="cuda", generator=g).half()
    for cross in (True, False):
        def triple():
            f3 = hip.mlp_fan(x, fq, bqkv, ws)
            v3 = f3.reshape(nimg * h, w, 3 * C)
            o = hip.attention(v3[..., :C], v3[..., C:2 * C], v3[..., 2 * C:], heads, swap_halves=cross).reshape(nimg, h, w, C)
            return hip.mlp_chain(o, st, res=x, res_stage=0, carry=True, frag=True)
        t_new = timeit_graph(lambda: hip.row_attn(x, heads, cross, weights, vectors), 20, 3)
        t_old = timeit_graph(triple, 20, 3)
        fl = 2.0 * nimg * h * w * C * C * 6 + 4.0 * nimg * h * w * w * C
        print(f"{name:14s} {'cross' if cross else 'self '}: row_attn {t_new:7.1f} us ({fl / t_new / 1e6:6.1f} TF/s)   fan + attention + chain {t_old:7.1f} us", flush=True)
