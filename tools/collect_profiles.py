#!/usr/bin/env python3
"""Copy the summaries of a tools/gpu_profile_rNN.sh pass from gpurun_out/<tag>/ into profiles/<dst>/ (tracked) and derive the K1 PMC
traffic JSON + the rocprofv3-vs-bench agreement line.     python tools/collect_profiles.py r03p [r03]"""
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
for f in ("bench_N1.json", "bench_N1_banded.json", "kbench.txt", "convbench_3x3_cold.txt", "attnbench.txt", "k1_timeline.txt",
          "frag_timeline.txt", "fusionbench.txt", "layer_ab_frag.txt", "layer_trace_eager.txt", "parity_c1_c3_c2.txt", "configs_all_models.txt",
          "pytest_gpu.txt", "attnbench_kt4.txt", "convbench_3x3_hot.txt", "k1_store_path.txt", "clock_probe.txt", "layer_ab_frag_pw.txt",
          "layer_ab_frag_aux_pw.txt", "parity_tables.txt", "parity_c4_c5_vs_reference.txt", "ab_k1_own_ln_dense.json", "ab_frag_pw32.json",
          "ab_frag_aux_pw32.json", "ab_k2_notri.json", "ab_default.json", "ab_no_fan_ws.json", "ab_no_pw_ws.json", "ab_no_chain_ws.json",
          "ab_gru_separate.json", "ab_round3_all_off.json", "chainbench.txt", "chainbench_no_ws.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f))
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
lines.append(f"sum of all kernel durations in the rocprofv3 run: {tot / 1e6:.2f} ms")
open(os.path.join(dst, "k1_bench_vs_rocprof.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
print(sorted(os.listdir(dst)))
