#!/bin/bash
# Round-3 measurement pass on the GPU box (via gpurun): everything lands under gpurun_out/<tag>/; tools/collect_profiles.py then
# copies the summaries worth keeping into profiles/r03/.   usage: bash tools/gpu_profile_r03.sh r03p
set -u
TAG=${1:-r03p}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# 0. the whole -m gpu suite
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v amdgpu.ids | tail -15 > $OUT/pytest_gpu.txt; echo "pytest rc=${PIPESTATUS[0]}"; tail -3 $OUT/pytest_gpu.txt
# 1. the bench line exactly as the driver runs it
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_N1.json 2> $OUT/bench_N1.err; echo "bench rc=$?"
# 2. rocprofv3 kernel stats of the same command (no CPU baseline: that leg is host-only)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/prof_bench.log 2>&1
echo "rocprof bench rc=$?"
# 3. K1 alone (the shipped form): stats + PMC passes (separate runs per counter)
for CASE in c3 c2; do
  timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_k1_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/prof_k1_$CASE.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -T -f csv -d $OUT/pmc_fetch_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/pmc_fetch_$CASE.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -T -f csv -d $OUT/pmc_write_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/pmc_write_$CASE.log 2>&1
done
cd $R
# 4. micro-benchmarks and timelines
timeout 600 python tools/kbench.py --iters 30 2>&1 | grep -v amdgpu.ids > $OUT/kbench.txt
timeout 300 python tools/attnbench.py 2>&1 | grep -v amdgpu.ids > $OUT/attnbench.txt
S2M2_LIB_SUFFIX=_kt4 timeout 300 python tools/attnbench.py 2>&1 | grep -v amdgpu.ids > $OUT/attnbench_kt4.txt
timeout 600 python tools/convbench.py --cold --only 3x --tiles 26,24 2>&1 | grep -v amdgpu.ids > $OUT/convbench_3x3_cold.txt
timeout 600 python tools/convbench.py --only 3x --tiles 26,24 2>&1 | grep -v amdgpu.ids > $OUT/convbench_3x3_hot.txt
timeout 200 python tools/k1_store_path.py c3 2>&1 | grep -v amdgpu.ids > $OUT/k1_store_path.txt
S2M2_LIB_SUFFIX=_k1trace timeout 120 python tools/k1_trace.py c3 --prenorm --aligned 2>&1 | grep -v amdgpu.ids > $OUT/k1_timeline.txt
timeout 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/clock_probe.txt
timeout 300 python tools/layer_trace.py --iters 3 2>&1 | grep -v amdgpu.ids > $OUT/layer_trace_eager.txt
timeout 300 python tools/chainbench.py 2>&1 | grep -v amdgpu.ids > $OUT/chainbench.txt
S2M2_CHAIN_WS=0 timeout 300 python tools/chainbench.py 2>&1 | grep -v amdgpu.ids > $OUT/chainbench_no_ws.txt
# 5. parity tables at the BASELINE sizes + the other configurations through the drop-in module
timeout 900 python tools/parity_report.py $OUT/parity_tables.txt > /dev/null 2>&1
timeout 600 python tools/parity_report_big.py $OUT/parity_c4_c5_vs_reference.txt > /dev/null 2>&1
timeout 900 python tools/configs_run.py 2>&1 | grep -v amdgpu.ids > $OUT/configs_all_models.txt
# 6. same-box A/B of this round's switches, end to end
bench() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null; }
bench S2M2_FUSE_K1LN=0 S2M2_CV_ALIGNED=0 > $OUT/ab_k1_own_ln_dense.json
bench S2M2_FRAG_PW=32 S2M2_FRAG_AUX_PW=32 > $OUT/ab_frag_pw32.json
bench S2M2_FRAG_AUX_PW=32 > $OUT/ab_frag_aux_pw32.json
bench S2M2_K2_TRI=0 > $OUT/ab_k2_notri.json
bench S2M2_FAN_WS=0 > $OUT/ab_no_fan_ws.json
bench S2M2_PW_WS=0 > $OUT/ab_no_pw_ws.json
bench S2M2_CHAIN_WS=0 > $OUT/ab_no_chain_ws.json
bench S2M2_FUSE_GRU=0 > $OUT/ab_gru_separate.json
bench S2M2_FAN_WS=0 S2M2_PW_WS=0 S2M2_CHAIN_WS=0 S2M2_FUSE_GRU=0 S2M2_FRAG_PW=32 S2M2_FRAG_AUX_PW=32 S2M2_K2_TRI=0 S2M2_FUSE_K1LN=0 S2M2_CV_ALIGNED=0 > $OUT/ab_round3_all_off.json
bench A=1 > $OUT/ab_default.json
python - <<PY
import json, glob
for n in ["$OUT/bench_N1.json"] + sorted(glob.glob("$OUT/ab_*.json")):
    try:
        d = json.load(open(n))
        print(n.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms  K1", round(d["roofline"]["avg_launch_us"], 2), "us frac", round(d["roofline"]["frac"], 3),
              "attn", round(d["roofline_attention"]["frac"], 4), "fwd", round(d["forward"]["frac_of_mfma_peak"], 4), d.get("secondary", {}).get("value"), d.get("secondary_batched", {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
du -sh $OUT
