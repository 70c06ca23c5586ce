#!/bin/bash
# Round-3 second GPU pass: whole -m gpu suite (no -x), K1 variants same-box, K1 timelines of both variants, conv 4x40 A/B.
set -u
TAG=${1:-r03b}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 -s 2>&1 | grep -v amdgpu.ids > $OUT/pytest_gpu_full.txt; echo "pytest rc=${PIPESTATUS[0]}"
grep -E "passed|failed|FAILED|ERROR" $OUT/pytest_gpu_full.txt | tail -40
for V in "1 1" "0 0" "1 0"; do
  set -- $V
  S2M2_FUSE_K1LN=$1 S2M2_CV_ALIGNED=$2 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_k1_fold$1_align$2.json 2>/dev/null
done
S2M2_FRAG_PW=32 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_pw32.json 2>/dev/null
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_default.json 2>/dev/null
python - <<PY
import json, glob
for n in sorted(glob.glob("$OUT/bench_*.json")):
    try:
        d = json.load(open(n))
        print(n.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms  K1", round(d["roofline"]["avg_launch_us"], 2), "us frac", round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(n, "failed", e)
PY
S2M2_LIB_SUFFIX=_k1trace timeout 120 python tools/k1_trace.py c3 2>&1 | grep -v amdgpu.ids > $OUT/k1_timeline_ln.txt
S2M2_LIB_SUFFIX=_k1trace timeout 120 python tools/k1_trace.py c3 --prenorm 2>&1 | grep -v amdgpu.ids > $OUT/k1_timeline_prenorm.txt
S2M2_LIB_SUFFIX=_k1trace timeout 120 python tools/k1_trace.py c3 --prenorm --aligned 2>&1 | grep -v amdgpu.ids > $OUT/k1_timeline_prenorm_aligned.txt
cat $OUT/k1_timeline_ln.txt $OUT/k1_timeline_prenorm.txt $OUT/k1_timeline_prenorm_aligned.txt
