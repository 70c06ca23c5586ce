#!/bin/bash
# Round-3 third GPU pass: the tests that failed in pass b, K1 store path / NT stores, K2 with the LDS triangle, layer trace.
set -u
TAG=${1:-r03c}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_dispinit.py tests/test_reference_golden_big.py tests/test_hip_utils.py tests/test_hip_lds_poison.py "tests/test_hip_parity_baseline.py::test_fp16_640x480_sharp_matches_free_running" "tests/test_hip_parity_baseline.py::test_fp32_1216x1024_every_stage" tests/test_hip_e2e.py -m gpu -q --timeout 900 -s 2>&1 | grep -v amdgpu.ids > $OUT/pytest_gpu_part.txt; echo "pytest rc=${PIPESTATUS[0]}"
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu_part.txt | tail -20
timeout 200 python tools/k1_store_path.py c3 2>&1 | grep -v amdgpu.ids > $OUT/k1_store_path.txt
S2M2_K1_NT=1 timeout 200 python tools/k1_store_path.py c3 2>&1 | grep -v amdgpu.ids | head -5 >> $OUT/k1_store_path.txt
cat $OUT/k1_store_path.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_default.json 2>/dev/null
S2M2_K1_NT=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_k1nt.json 2>/dev/null
S2M2_K9_XCD=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_k9_noxcd.json 2>/dev/null
S2M2_K2_TRI=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_k2_notri.json 2>/dev/null
python - <<PY
import json, glob
for n in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(n))
        print(n.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms  K1", round(d["roofline"]["avg_launch_us"], 2), "us frac", round(d["roofline"]["frac"], 3), d.get("secondary", {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
timeout 300 python tools/kbench.py --iters 30 2>&1 | grep -v amdgpu.ids > $OUT/kbench.txt; cat $OUT/kbench.txt
S2M2_K2_TRI=0 timeout 300 python tools/kbench.py --iters 30 2>&1 | grep -v amdgpu.ids > $OUT/kbench_k2_notri.txt; grep -i "K2\|sinkhorn" $OUT/kbench_k2_notri.txt
timeout 300 python tools/layer_trace.py --iters 3 2>&1 | grep -v amdgpu.ids > $OUT/layer_trace_eager.txt; head -70 $OUT/layer_trace_eager.txt
