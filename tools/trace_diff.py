#!/usr/bin/env python3
"""Per-layer difference of two tools/layer_trace.py outputs (A/B of a kernel switch inside the real pipeline).
    python tools/trace_diff.py new.txt old.txt"""
import re
import sys


def load(p):
    d = {}
    for ln in open(p):
        m = re.match(r"(\S+)\s+(.*?)\s+calls/pass\s+(\d+)\s+avg\s+([\d.]+) us\s+total\s+([\d.]+) us", ln)
        if m:
            d[(m.group(1), m.group(2).replace(" frag", ""))] = (int(m.group(3)), float(m.group(4)), float(m.group(5)), " frag" in m.group(2))
    return d


a, b = load(sys.argv[1]), load(sys.argv[2])
rows = [(b[k][2] - v[2], k, v, b[k]) for k, v in a.items() if k in b and abs(b[k][2] - v[2]) > 3]
tot = 0.0
for d, k, v, w in sorted(rows, key=lambda r: r[0]):
    print(f"{k[0]:14s} {k[1]:56s} x{v[0]:2d}  new {v[1]:7.1f}  old {w[1]:7.1f} us   saved/pass {d:8.1f} us {'(frag)' if v[3] else ''}")
    tot += d
print("total saved per pass: %.1f us" % tot)
print(open(sys.argv[1]).readline().strip())
print(open(sys.argv[2]).readline().strip())
