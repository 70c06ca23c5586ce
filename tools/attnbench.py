#!/usr/bin/env python3
"""Time s2m2_attention against torch SDPA on the hot-path shapes (fp16)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip
from s2m2_amd.engine import pe_tables
from tools.kbench import timeit_graph
timeit = lambda fn, n: timeit_graph(fn, 20, 3)
SHAPES = [("L0 self 1-D", 512, 1, 304, 128, False, None), ("L0 cross 1-D", 512, 1, 304, 128, True, None), ("L1 1-D", 256, 2, 152, 64, False, None),
          ("L2 1-D", 128, 4, 76, 64, False, None), ("L3 2-D self", 2, 8, 1216, 32, False, None), ("L3 2-D cross", 2, 8, 1216, 32, True, None),
          ("refiner 2-D", 1, 8, 1216, 32, False, None), ("global ref 2-D", 1, 8, 1216, 16, False, None), ("pyramid PE", 2, 8, 1216, 32, False, (32, 38)),
          ("M L0 1-D", 512, 1, 304, 192, False, None), ("L L0 1-D", 512, 1, 304, 256, False, None), ("XL L0 1-D", 1024, 1, 608, 384, False, None)]
ONLY = os.environ.get("ATTNBENCH_ONLY", "")          # substring filter on the shape names
for name, nb, h, N, d, swap, grid in SHAPES:
    if ONLY and ONLY not in name:
        continue
    C = h * d
    qkv = torch.randn(nb, N, 3 * C, device="cuda").half()
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    pe = None
    if grid:
        px, py = pe_tables(grid[0], grid[1], "cuda")
        pe = (px, py, grid[1], grid[0])
    t = timeit(lambda: hip.attention(q, k, v, h, swap_halves=swap, pe=pe), 20)
    sp = lambda t_: t_.reshape(nb, N, h, d).transpose(1, 2)
    tr = timeit(lambda: F.scaled_dot_product_attention(sp(q), sp(k), sp(v)), 20)
    fl = 4.0 * nb * h * N * N * d
    print(f"{name:16s} nb={nb:4d} h={h} N={N:5d} d={d:4d}: hip {t:8.1f} us ({fl / t / 1e6:6.1f} TF/s)   torch sdpa {tr:8.1f} us", flush=True)
