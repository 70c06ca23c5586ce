#!/usr/bin/env python3
"""MFMA utilisation of s2m2_attention per launch shape from a rocprofv3 --pmc counter_collection.csv (tools/pmc_attn.sh):
util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles), kernel cycles = SQ_BUSY_CYCLES / 32 shader engines."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
disp = collections.OrderedDict()
for r in rows:
    if "attention_kernel" not in r["Kernel_Name"]:
        continue
    d = disp.setdefault(r["Dispatch_Id"], {"name": r["Kernel_Name"], "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(list)
for d in disp.values():
    n = d["name"]
    agg[(n[n.index("AttnCfg"):][:40], d["grid"], d["wg"])].append(d)
print("# s2m2_attention on MI355X, shapes of tools/attnbench.py (fp16); MFMA util vs 1024 SIMDs, dense peak 2.5 PFLOP/s")
print(f"{'kernel config':42s} {'grid':>8s} {'wg':>4s} {'launches':>8s} {'kernel_cycles':>13s} {'mfma_busy':>11s} {'MFMA util':>9s} {'VALU/wave':>9s} {'wait/wave':>9s}")
for k, v in agg.items():
    m = sum(x["SQ_VALU_MFMA_BUSY_CYCLES"] for x in v) / len(v)
    b = sum(x["SQ_BUSY_CYCLES"] for x in v) / len(v) / 32
    wc = sum(x["SQ_WAVE_CYCLES"] for x in v) / len(v)
    valu = sum(x["SQ_ACTIVE_INST_VALU"] for x in v) / len(v)
    wa = sum(x["SQ_WAIT_ANY"] for x in v) / len(v)
    print(f"{k[0]:42s} {k[1]:>8s} {k[2]:>4s} {len(v):8d} {b:13.0f} {m:11.3g} {m / (1024 * b):9.3f} {valu / wc:9.2f} {wa / wc:9.2f}")
