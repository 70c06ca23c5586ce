#!/bin/bash
# Round-3 first GPU pass: the whole -m gpu suite, the bench line, and a same-box A/B of the K1 changes.   usage: bash tools/gpu_r03a.sh <tag>
set -u
TAG=${1:-r03a}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 2>&1 | grep -v amdgpu.ids | tail -40 > $OUT/pytest_gpu.txt; echo "pytest rc=${PIPESTATUS[0]}"
tail -5 $OUT/pytest_gpu.txt
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_N1.json 2> $OUT/bench_N1.err; echo "bench rc=$?"
S2M2_FUSE_K1LN=0 S2M2_CV_ALIGNED=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_N1_k1_own_ln_dense.json 2>/dev/null
S2M2_FUSE_K1LN=1 S2M2_CV_ALIGNED=0 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_N1_k1_folded_dense.json 2>/dev/null
S2M2_FUSE_K1LN=0 S2M2_CV_ALIGNED=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/bench_N1_k1_own_ln_aligned.json 2>/dev/null
python - <<PY
import json
for n in ("bench_N1", "bench_N1_k1_own_ln_dense", "bench_N1_k1_folded_dense", "bench_N1_k1_own_ln_aligned"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
        print(n, round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms  K1", round(d["roofline"]["avg_launch_us"], 2), "us frac", round(d["roofline"]["frac"], 3),
              "| 640x480:", d.get("secondary", {}).get("value"))
    except Exception as e:
        print(n, "failed", e)
PY
timeout 300 python tools/clock_probe.py 2>&1 | grep -v amdgpu.ids > $OUT/clock_probe.txt; cat $OUT/clock_probe.txt
