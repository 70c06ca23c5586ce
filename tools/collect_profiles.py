#!/usr/bin/env python3
"""Copy the summaries of a tools/gpu_pass.sh pass from gpurun_out/<tag>/ into profiles/<dst>/ (tracked) and derive the K1 PMC
traffic JSON + the rocprofv3-vs-bench agreement line.     python tools/collect_profiles.py r04z r04"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
dtag = sys.argv[2] if len(sys.argv) > 2 else tag
src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles", dtag)
os.makedirs(dst, exist_ok=True)
for f in ("bench_N1.json", "kbench.txt", "attnbench.txt", "pwbench.txt", "layer_trace_eager.txt", "parity_tables.txt", "configs_all_models.txt",
          "pytest_gpu.txt", "k1_store_modes.txt", "head.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
hl = os.path.join(ROOT, "gpurun_out", "fp16_headline")                      # written by tests/test_fp16_headline.py
if os.path.isdir(hl):
    for f in sorted(os.listdir(hl)):
        shutil.copy(os.path.join(hl, f), os.path.join(dst, "fp16_headline_" + f))
copies = {"prof_bench/bench_kernel_stats.csv": "bench_c3_S_fp16_kernel_stats.csv", "prof_k1_c3/k1_kernel_stats.csv": "k1_only_c3_kernel_stats.csv",
          "prof_k1_c2/k1_kernel_stats.csv": "k1_only_c2_kernel_stats.csv"}
for a, b in copies.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
for case in ("c3", "c2"):
    f, w = (os.path.join(src, f"pmc_{k}_{case}", "k1_counter_collection.csv") for k in ("fetch", "write"))
    if os.path.exists(f) and os.path.exists(w):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f, w], capture_output=True, text=True).stdout
        open(os.path.join(dst, f"k1_{case}_fp16_pmc.json"), "w").write(out)
# agreement of the bench line's K1 duration (HIP events on the dispatch) with rocprofv3's average for the same command
bench = json.load(open(os.path.join(dst, "bench_N1.json")))
rows = list(csv.DictReader(open(os.path.join(dst, "bench_c3_S_fp16_kernel_stats.csv"))))
k1 = [r for r in rows if "ln_corr" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lines = [f"bench.py line ({tag}/bench_N1.json): K1 avg {bench['roofline']['avg_launch_us']:.2f} us over {bench['roofline']['launches_timed']} launches "
         f"(start/stop HIP events attached to each dispatch) -> {bench['roofline']['achieved']:.0f} GB/s = {bench['roofline']['frac']:.3f} of 8 TB/s"]
for r in k1:
    lines.append(f"rocprofv3 --kernel-trace --stats of the same command ({tag}/bench_c3_S_fp16_kernel_stats.csv): {r['Name'][:60]}... calls {r['Calls']} "
                 f"avg {float(r['AverageNs']) / 1e3:.2f} us min {float(r['MinNs']) / 1e3:.2f} max {float(r['MaxNs']) / 1e3:.2f}")
# the stats average mixes the forward's launches with the back-to-back launches of the bench line's `both_variants` leg: split them by
# what ran before (kernel trace of the same rocprofv3 run)
trace = os.path.join(src, "prof_bench", "bench_kernel_trace.csv")
if os.path.exists(trace):
    tr = sorted(csv.DictReader(open(trace)), key=lambda r: int(r["Start_Timestamp"]))
    infwd, b2b, prev = [], [], ""
    for r in tr:
        if "ln_corr" in r["Kernel_Name"]:
            (b2b if "ln_corr" in prev else infwd).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        prev = r["Kernel_Name"]
    if infwd:
        lines.append(f"  same run, kernel trace: {len(infwd)} K1 launches INSIDE a forward (previous kernel is not K1) avg {sum(infwd) / len(infwd):.2f} us, "
                     f"the last 20 (the timed steps) {sum(infwd[-20:]) / len(infwd[-20:]):.2f} us; {len(b2b)} back-to-back launches (both_variants leg) "
                     f"avg {sum(b2b) / max(1, len(b2b)):.2f} us")
lines.append(f"sum of all kernel durations in the rocprofv3 run: {tot / 1e6:.2f} ms")
open(os.path.join(dst, "k1_bench_vs_rocprof.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
print(sorted(os.listdir(dst)))
