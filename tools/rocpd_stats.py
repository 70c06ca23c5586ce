#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace): per-kernel calls / total / avg / min / max duration.

    python tools/rocpd_stats.py <results.db> [top_n] > profiles/<name>_kernel_stats.txt
"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, (end - start) from kernels").fetchall()
agg = {}
for name, d in rows:
    short = re.sub(r"\(.*", "", name)
    short = re.sub(r"^void ", "", short)
    a = agg.setdefault(short, [0, 0, 1 << 62, 0])
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values())
print(f"# {sys.argv[1]}: {len(rows)} dispatches, {len(agg)} kernels, total GPU kernel time {total / 1e6:.3f} ms")
print(f"{'calls':>7} {'total_ms':>10} {'%':>6} {'avg_us':>9} {'min_us':>9} {'max_us':>9}  kernel")
for name, (n, t, mn, mx) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n:7d} {t / 1e6:10.3f} {100 * t / total:6.2f} {t / n / 1e3:9.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f}  {name[:140]}")
