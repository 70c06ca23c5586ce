#!/usr/bin/env python3
"""One steady-state forward out of a rocprofv3 --kernel-trace CSV: every launch in stream order with its grid, resources, duration and the gap to the
previous kernel's end; then per-family totals.  The forwards are delimited by the image_prep kernel (the first launch of a forward).

    python tools/trace_forward.py <bench_kernel_trace.csv> [which=-2] [--all]      (which: index of the forward, default the last but one)
"""
import csv
import re
import sys

path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else -2
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    n = re.sub(r"^_ZN4s2m2\d+", "", n)
    m = re.match(r"([a-z_0-9]+_kernel)I(.*?)E+v", n)
    if m:
        args = re.findall(r"Li(\d+)|Lb(\d)", m.group(2))
        return m.group(1).replace("_kernel", "") + "<" + ",".join(a or b for a, b in args) + ">"
    return re.sub(r"\(.*", "", n)[:60]


starts = [i for i, r in enumerate(rows) if "image_prep" in r["Kernel_Name"]]
if len(starts) < 3:
    sys.exit("fewer than three forwards in the trace")
a = starts[which]
b = starts[which + 1] if which + 1 < 0 or which + 1 < len(starts) else len(rows)
if which == -1:
    b = len(rows)
fw = rows[a:b]
t0 = int(fw[0]["Start_Timestamp"])
prev_end = t0
tot = 0
fam = {}
print(f"# forward {which} of {len(starts)}: {len(fw)} launches, wall {(int(fw[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
print(f"{'#':>4} {'t_us':>8} {'dur_us':>8} {'gap_us':>7} {'blocks':>7} {'thr':>4} {'lds':>7} {'vgpr':>4} {'agpr':>4}  kernel")
for i, r in enumerate(fw):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    blocks = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(wg, 1)
    nm = short(r["Kernel_Name"])
    d = (e - s) / 1e3
    tot += d
    f = fam.setdefault(nm, [0, 0.0, 0])
    f[0] += 1
    f[1] += d
    f[2] += blocks
    if "--all" in sys.argv:
        print(f"{i:4d} {(s - t0) / 1e3:8.1f} {d:8.2f} {(s - prev_end) / 1e3:7.2f} {blocks:7d} {wg:4d} {int(r['LDS_Block_Size']):7d} {int(r['VGPR_Count']):4d} {int(r['Accum_VGPR_Count']):4d}  {nm}")
    prev_end = e
print(f"# sum of kernel durations {tot:.1f} us")
print(f"{'calls':>5} {'total_us':>9} {'%':>6} {'avg_us':>8} {'avg_blocks':>10}  kernel")
for nm, (n, t, bl) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:5d} {t:9.1f} {100 * t / tot:6.2f} {t / n:8.2f} {bl / n:10.0f}  {nm}")
