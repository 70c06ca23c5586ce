#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per launch) and apply the gfx950 HBM corrections of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reports 1/2 of a wide coalesced read stream
(x2), checked here against a device copy of a known byte count in the same run (tools/k1_only.py); WRITE_SIZE is exact on it.

    python tools/pmc_summary.py <fetch.csv> <write.csv> [kernel-substring] > profiles/rNN/<name>_pmc.json
"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def k1_source_hash():
    """the sources K1 is compiled from (bench.py quotes a summary only while this still matches: bench.k1_source_hash)"""
    h = hashlib.sha256()
    for f in ("ln_corr.hip", "common.h"):
        h.update(open(os.path.join(ROOT, "s2m2_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    pat = sys.argv[3] if len(sys.argv) > 3 else "ln_corr"
    out = {"units": "bytes per launch", "fetch_correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read half counting)",
           "write_correction": "WRITE_SIZE KiB x 1024"}
    cal_f = [v for k, v in fetch.items() if "copyBuffer" in k]
    cal_w = [v for k, v in write.items() if "copyBuffer" in k]
    if cal_f and cal_w:
        out["calibration_copy"] = {"fetch_bytes_corrected": cal_f[0][0] * 1024 * 2, "write_bytes": cal_w[0][0] * 1024,
                                   "note": "hipMemcpy DtoD of a known size issued by tools/k1_only.py (--calib-mib, default 512 MiB)"}
    for k, (v, n) in fetch.items():
        if pat in k:
            w, nw = write.get(k, (float("nan"), 0))
            out["kernel"] = k
            out["launches"] = n
            out["fetch_bytes"] = v * 1024 * 2
            out["write_bytes"] = w * 1024
            out["traffic_bytes"] = v * 1024 * 2 + w * 1024
    if pat == "ln_corr":
        out["k1_source_sha256_16"] = k1_source_hash()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
