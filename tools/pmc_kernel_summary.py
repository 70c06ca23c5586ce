#!/usr/bin/env python3
"""Per kernel (name contains argv[1]): average duration from a rocprofv3 kernel trace and average counter values per dispatch from one or more
--pmc counter_collection.csv files (tools/pmc_kernel.sh), plus the usual ratios.  SQ_* cycle counters are in quad-cycles per the guide, MFMA busy in cycles."""
import collections
import csv
import re
import sys

sub, files = sys.argv[1], sys.argv[2:]


def short(n):
    return re.sub(r"\(.*", "", n)[:70]


dur = collections.defaultdict(list)
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for f in files:
    rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    if "Counter_Name" in rows[0]:
        per = collections.defaultdict(dict)
        for r in rows:
            if sub in r["Kernel_Name"]:
                per[(r["Dispatch_Id"], short(r["Kernel_Name"]), r["Grid_Size"], r["Workgroup_Size"])][r["Counter_Name"]] = float(r["Counter_Value"])
        for (_, n, g, w), c in per.items():
            for k, v in c.items():
                ctr[(n, g, w)][k].append(v)
    else:
        for r in rows:
            if sub in r["Kernel_Name"]:
                g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                w = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
                dur[(short(r["Kernel_Name"]), str(g), str(w))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for key in sorted(set(dur) | set(ctr)):
    n, g, w = key
    d = dur.get(key, [])
    c = {k: sum(v) / len(v) for k, v in ctr.get(key, {}).items()}
    print(f"== {n}  grid {g} wg {w}: {len(d)} launches, avg {sum(d) / max(len(d), 1):.2f} us")
    for k in sorted(c):
        print(f"   {k:28s} {c[k]:14.4g}")
    if "SQ_BUSY_CYCLES" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        kc = c["SQ_BUSY_CYCLES"] / 32
        print(f"   kernel cycles {kc:.0f}; MFMA util {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * kc):.3f} of 1024 SIMDs")
    if "SQ_WAVE_CYCLES" in c:
        wc = c["SQ_WAVE_CYCLES"]
        print("   per wave-cycle: " + "  ".join(f"{k[3:].lower()} {c[k] / wc:.2f}" for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU") if k in c))
    if "SQ_LDS_IDX_ACTIVE" in c:
        print(f"   LDS: bank-conflict cycles / active cycles {c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c['SQ_LDS_IDX_ACTIVE'], 1):.3f}; "
              f"LDS active / GUI active {c['SQ_LDS_IDX_ACTIVE'] / max(c.get('GRBM_GUI_ACTIVE', 1), 1):.3f} (summed over CUs)")
