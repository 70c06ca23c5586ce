#!/usr/bin/env python3
"""Per-kernel VALU / MFMA pipe occupancy from the rocprofv3 --pmc passes of tools/pmc_valu.sh.
kernel cycles = SQ_BUSY_CYCLES / 32 (shader engines x ...: the normalisation tools/pmc_attn_summary.py calibrated); a wave64 VALU
instruction occupies its SIMD for 4 cycles (transcendentals 16), so VALU issue utilisation >= 4 * SQ_INSTS_VALU / (1024 SIMDs x kernel cycles)."""
import collections
import csv
import glob
import os
import sys

out = sys.argv[1]
for variant in ("ws", "stream"):
    agg = collections.OrderedDict()
    for sub in ("sq", "lds"):
        for f in glob.glob(os.path.join(out, f"{sub}_{variant}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                n = r["Kernel_Name"]
                key = "K9 chain" if "mlp_chain" in n else "K10 fusion" if "feature_fusion" in n else "K2 sinkhorn" if "sinkhorn" in n else None
                if key is None:
                    continue
                agg.setdefault(key, collections.defaultdict(list))[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== K9 form: {'weights-stationary' if variant == 'ws' else 'streaming (S2M2_CHAIN_WS=0)'}")
    print(f"{'kernel':14s}{'kernel cycles':>14s}{'VALU insts/SIMD':>16s}{'VALU issue util':>16s}{'trans share':>12s}{'MFMA util':>10s}{'LDS insts/SIMD':>15s}{'wave busy':>10s}")
    for k, c in agg.items():
        m = {n: sum(v) / len(v) for n, v in c.items()}
        cyc = m.get("SQ_BUSY_CYCLES", 0) / 32
        if cyc <= 0:
            continue
        iv = m.get("SQ_INSTS_VALU", 0) / 1024
        tr = m.get("SQ_INSTS_VALU_TRANS", 0) / 1024
        util = (4 * (iv - tr) + 16 * tr) / cyc if tr else 4 * iv / cyc
        print(f"{k:14s}{cyc:14.0f}{iv:16.0f}{util:16.3f}{(tr / iv if iv else 0):12.3f}{m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * cyc):10.3f}"
              f"{m.get('SQ_INSTS_LDS', 0) / 1024:15.0f}{m.get('SQ_ACTIVE_INST_ANY', 0) / max(1.0, m.get('SQ_WAVE_CYCLES', 1)):10.2f}")
