#!/bin/bash
# Round-3 GPU pass h: fan-only WS (QKV), WS chain with deeper prefetch: tests + A/B.
set -u
TAG=${1:-r03h}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_chain.py tests/test_hip_e2e.py tests/test_hip_lds_poison.py tests/test_hip_dispinit.py tests/test_hip_attention.py "tests/test_hip_parity_baseline.py::test_fp16_640x480_pinned_to_autocast_emulation" tests/test_fp16_reference_autocast.py -m gpu -q --timeout 900 2>&1 | grep -v amdgpu.ids > $OUT/pytest_gpu_part.txt; echo "pytest rc=${PIPESTATUS[0]}"
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu_part.txt | tail -20
timeout 300 python tools/chainbench.py 2>&1 | grep -v amdgpu.ids | grep "1/4\|1/8" > $OUT/chainbench.txt; cat $OUT/chainbench.txt
bench() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null; }
bench A=1 > $OUT/bench_default_a.json
bench S2M2_FAN_WS=0 > $OUT/bench_no_fanws_a.json
bench A=2 > $OUT/bench_default_b.json
bench S2M2_FAN_WS=0 > $OUT/bench_no_fanws_b.json
bench S2M2_FAN_WS=0 S2M2_CHAIN_WS=0 > $OUT/bench_no_ws_at_all.json
python - <<PY
import json, glob
for n in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(n))
        print(n.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms  K1", round(d["roofline"]["avg_launch_us"], 2), "us frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(n, "failed", e)
PY
timeout 300 python tools/layer_trace.py --iters 3 2>&1 | grep -v amdgpu.ids | grep "mlp_fan\|mlp_chain\|k1x1 cin=128->384" > $OUT/layer_trace_chain.txt; cat $OUT/layer_trace_chain.txt
