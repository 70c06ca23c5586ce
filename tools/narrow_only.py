#!/usr/bin/env python3
"""Workload for rocprofv3 --pmc: every K12 layer shape of the S model at 1216x1024 launched a few times, eagerly (tools/narrow_pmc.sh)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from s2m2_amd import hip, pack  # noqa: E402
from tools.narrowbench import LAYERS  # noqa: E402

for name, k, stride, cs, cout, act, shp in LAYERS:
    g = torch.Generator(device="cuda").manual_seed(1)
    cin = sum(cs)
    srcs = [torch.randn(*shp, c, device="cuda", generator=g).half() for c in cs]
    w = (torch.randn(cout, cin, k, k, device="cuda", generator=g) / math.sqrt(cin * k * k)).half()
    wf = pack.narrow_frag(pack.pack_conv(w, torch.float16, [(c, c) for c in cs]), k * k)
    bp = pack.pack_bias(torch.randn(cout, device="cuda", generator=g), cout)
    for _ in range(4):
        hip.conv_narrow(srcs, wf, bp, k, k, cout, stride=stride, act=act)
    torch.cuda.synchronize()
    print(name, flush=True)
