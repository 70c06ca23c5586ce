#!/bin/bash
# Round-3 GPU pass o: plain 1x1 C->C layers and the LayerNorm-output launch on the direct K9 form -- e2e tests, end-to-end A/B.
set -u
TAG=${1:-r03o}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_e2e.py tests/test_hip_dispinit.py -m gpu -x -q -k "forward or e2e or graph or either_form" 2>&1 | tail -3
for rep in 1 2; do
  for v in "S2M2_PW_DIRECT=0 S2M2_CHAIN_DIRECT_LN=0" "S2M2_PW_DIRECT=1 S2M2_CHAIN_DIRECT_LN=0" "S2M2_PW_DIRECT=1 S2M2_CHAIN_DIRECT_LN=1"; do
    n=$(echo "$v" | tr ' =' '__')
    env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/ab_${n}_$rep.json 2>/dev/null; echo "$v rep=$rep rc=$?"
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms", "K1", round(d["roofline"]["avg_launch_us"], 2), round(d["roofline"]["frac"], 3))
PY
