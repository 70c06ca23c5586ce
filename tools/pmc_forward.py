#!/usr/bin/env python3
"""Which unit bounds each kernel of the forward: per kernel name (template arguments kept, argument list cut) the launches, the time, and from one
rocprofv3 --pmc pass (SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY) the share of the
SIMD cycles with the matrix pipe busy and with a VALU instruction issuing (SQ_ACTIVE_INST_* count quad-cycles, MI355X_MICROARCH.md), the VALU
instructions per wave and the share of wave-cycles spent waiting.
    python tools/pmc_forward.py <kernel_trace.csv> <counter_collection.csv>     (tools/pmc_forward.sh makes both)"""
import collections
import csv
import re
import sys

trace, pmc = sys.argv[1], sys.argv[2]


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    n = n.replace("s2m2::", "")
    n = re.sub(r"(ConvCfg\w*|FusionDirectCfg|ChainCfg\w*|AttnCfg\w*|NarrowCfg|PxCfg|PwCfg\w*|CbCfg)<", "<", n)
    return n[:86]


dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
per = collections.defaultdict(dict)
for r in csv.DictReader(open(pmc)):
    per[(r["Dispatch_Id"], short(r["Kernel_Name"]), int(r["Grid_Size"]), int(r["Workgroup_Size"]))][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for (_, n, g, w), c in per.items():
    a = agg[n]
    a["n"] += 1
    a["waves"] += g / 64
    for k, v in c.items():
        a[k] += v
tot = sum(sum(v) for v in dur.values())
print(f"# {sum(len(v) for v in dur.values())} launches in the trace, {tot / 1e3:.2f} ms of kernel time; counters: one --pmc pass of the same command (eager launches)")
print(f"{'kernel':86s} {'launches':>8s} {'us total':>10s} {'share':>6s} {'MFMA busy':>9s} {'VALU busy':>9s} {'VALU/wave':>9s} {'waiting':>8s}")
for n, d in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    a = agg.get(n)
    line = f"{n:86s} {len(d):8d} {sum(d):10.1f} {sum(d) / tot:6.3f}"
    if a and a.get("SQ_BUSY_CYCLES"):
        simd_cycles = a["SQ_BUSY_CYCLES"] / 32 * 1024          # (as tools/pmc_kernel_summary.py: kernel cycles x 1024 SIMDs)
        line += f" {a.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles:9.3f} {4 * a.get('SQ_ACTIVE_INST_VALU', 0) / simd_cycles:9.3f}"
        line += f" {a.get('SQ_INSTS_VALU', 0) / max(a['waves'], 1):9.0f} {a.get('SQ_WAIT_ANY', 0) / max(a.get('SQ_WAVE_CYCLES', 1), 1):8.2f}"
    print(line)
