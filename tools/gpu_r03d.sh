#!/bin/bash
# Round-3 GPU pass d: streaming K1 (tests + timings + A/B switches), aux 160-px blocks A/B, the previously failing tests.
set -u
TAG=${1:-r03d}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_dispinit.py tests/test_hip_chain.py tests/test_hip_conv.py tests/test_hip_lds_poison.py "tests/test_hip_parity_baseline.py::test_fp16_640x480_sharp_matches_free_running" "tests/test_hip_parity_baseline.py::test_fp32_1216x1024_every_stage" tests/test_hip_e2e.py tests/test_fp16_reference_autocast.py -m gpu -q --timeout 900 2>&1 | grep -v amdgpu.ids > $OUT/pytest_gpu_part.txt; echo "pytest rc=${PIPESTATUS[0]}"
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu_part.txt | tail -20
timeout 200 python tools/k1_store_path.py c3 2>&1 | grep -v amdgpu.ids > $OUT/k1_store_path.txt; cat $OUT/k1_store_path.txt
for V in "S2M2_K1_NT=1" "S2M2_K1_STAGGER=1" "S2M2_K1_SPLIT=2" "S2M2_K1_SPLIT=2 S2M2_K1_STAGGER=1"; do
  echo "== $V"; env $V timeout 200 python tools/k1_store_path.py c3 2>&1 | grep "streaming"
done 2>&1 | tee $OUT/k1_stream_variants.txt
timeout 200 python tools/k1_store_path.py c2 2>&1 | grep -v amdgpu.ids | grep "K1 " > $OUT/k1_store_path_c2.txt; cat $OUT/k1_store_path_c2.txt
bench() { env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null; }
bench A=1 > $OUT/bench_default.json
bench S2M2_K1_STREAM=0 > $OUT/bench_k1_lds_form.json
bench S2M2_FRAG_AUX_PW=40 > $OUT/bench_aux_pw40.json
python - <<PY
import json, glob
for n in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(n))
        print(n.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms  K1", round(d["roofline"]["avg_launch_us"], 2), "us frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(n, "failed", e)
PY
