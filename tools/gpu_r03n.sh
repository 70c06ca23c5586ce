#!/bin/bash
# Round-3 GPU pass n: direct K9 at every row count (against the weights-stationary form at 1/4 resolution), end-to-end A/B.
set -u
TAG=${1:-r03n}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for rep in 1 2; do
  for v in "S2M2_CHAIN_DIRECT_MAX=40000" "S2M2_CHAIN_DIRECT_MAX=100000" "S2M2_CHAIN_DIRECT_MAX=200000"; do
    n=$(echo "$v" | tr ' =' '__')
    env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/ab_${n}_$rep.json 2>/dev/null; echo "$v rep=$rep rc=$?"
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms")
PY
