#!/bin/bash
# Round-3 GPU pass j: hybrid K1 (left tokens in fragment order) -- bit-for-bit tests, then an in-forward A/B against the row-major form on one box.
set -u
TAG=${1:-r03j}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_dispinit.py tests/test_hip_chain.py -m gpu -x -q -k "hybrid or either_form or fragment_order" 2>&1 | tail -4
for rep in 1 2; do
  for v in 0 1; do
    S2M2_K1_HYBRID=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/ab_hybrid${v}_$rep.json 2>/dev/null; echo "hybrid=$v rep=$rep rc=$?"
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_hybrid*.json")):
    d = json.load(open(f))
    r = d["roofline"]
    print(f.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms  K1", round(r.get("avg_launch_us", 0) or 0, 2), "us frac", round(r["frac"], 3), r.get("variant"))
PY
