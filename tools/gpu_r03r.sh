#!/bin/bash
# Round-3 GPU pass r: AvgPool2d(2) + 1x1 (the down_convs) as fan-out-only launches of the direct K9 form -- tests, end-to-end A/B.
set -u
TAG=${1:-r03r}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_hip_chain.py -m gpu -x -q -k "avgpool or fan_only" 2>&1 | tail -3
S2M2_POOL_DIRECT=1 timeout 200 python -m pytest tests/test_hip_e2e.py -m gpu -x -q -k "graph_replay or fp16_forward" 2>&1 | tail -2
for rep in 1 2; do
  for v in "S2M2_POOL_DIRECT=0" "S2M2_POOL_DIRECT=1"; do
    n=$(echo "$v" | tr ' =' '__')
    env $v timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/ab_${n}_$rep.json 2>/dev/null; echo "$v rep=$rep rc=$?"
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms")
PY
