#!/bin/bash
# Round-3 closing pass on the GPU box: the whole -m gpu suite, smoke(), the bench line as the driver runs it, rocprofv3 kernel stats of the same
# command and of K1 alone, the eager layer trace.   usage: bash tools/gpu_final_r03.sh r03z ; python tools/collect_profiles.py r03z r03
set -u
TAG=${1:-r03z}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -v amdgpu.ids | tail -15 > $OUT/pytest_gpu.txt; echo "pytest rc=${PIPESTATUS[0]}"; tail -3 $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_N1.json 2> $OUT/bench_N1.err; echo "bench rc=$?"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/prof_bench.log 2>&1
echo "rocprof bench rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_k1_c3 -o k1 -- python $R/tools/k1_only.py --case c3 > $OUT/prof_k1_c3.log 2>&1
cd $R
timeout 300 python tools/layer_trace.py --iters 3 2>&1 | grep -v amdgpu.ids > $OUT/layer_trace_eager.txt
python - <<PY
import json
d = json.load(open("$OUT/bench_N1.json"))
print("bench_N1", round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms  K1", round(d["roofline"]["avg_launch_us"], 2), "us frac", round(d["roofline"]["frac"], 3),
      "attn", round(d["roofline_attention"]["frac"], 4), "fwd", round(d["forward"]["frac_of_mfma_peak"], 4), d.get("secondary_640x480", {}).get("value"), d.get("secondary_batched", {}).get("value"),
      "cpu", d.get("cpu_baseline", {}).get("value"))
PY
rm -rf $OUT/prof_bench/*kernel_trace.csv $OUT/prof_k1_c3/*kernel_trace.csv
du -sh $OUT
