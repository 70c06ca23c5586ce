"""Dump key/shape lists of the reference ``S2M2.state_dict()`` for the four published model sizes.

Run in the build container only (needs /root/reference):
    python tests/golden/make_spec_fixtures.py
Writes tests/golden/state_dict_spec_{S,M,L,XL}.json.
"""
import json
import os
import sys

sys.path.insert(0, "/root/reference/src")
import torch  # noqa: E402
from s2m2.core.model.s2m2 import S2M2  # noqa: E402  (reference, read-only)

HERE = os.path.dirname(os.path.abspath(__file__))
CONFIGS = {"S": (128, 1), "M": (192, 2), "L": (256, 3), "XL": (384, 3)}

for name, (c, n) in CONFIGS.items():
    with torch.device("meta"):
        m = S2M2(c, 1, n, True, False, 3)
    spec = [[k, list(v.shape)] for k, v in m.state_dict().items()]
    nparam = sum(int(torch.Size(s).numel()) for _, s in spec)
    with open(os.path.join(HERE, f"state_dict_spec_{name}.json"), "w") as f:
        json.dump({"feature_channels": c, "num_transformer": n, "num_parameters": nparam, "entries": spec}, f)
    print(name, len(spec), nparam)
