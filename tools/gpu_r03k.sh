#!/bin/bash
# Round-3 GPU pass k: direct form of K9 (weights in fragment order) -- bit-for-bit tests, micro-benchmark at the short row counts, end-to-end A/B.
set -u
TAG=${1:-r03k}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_chain.py tests/test_hip_e2e.py tests/test_hip_lds_poison.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python tools/chainbench_direct.py > $OUT/chainbench_direct.txt 2>&1; echo "chainbench rc=$?"; cat $OUT/chainbench_direct.txt | grep -v amdgpu.ids
for rep in 1 2; do
  for v in "S2M2_CHAIN_DIRECT=0" "S2M2_CHAIN_DIRECT=1" "S2M2_CHAIN_DIRECT_MAX=40000"; do
    n=$(echo "$v" | tr ' =' '__')
    env $v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/ab_${n}_$rep.json 2>/dev/null; echo "$v rep=$rep rc=$?"
  done
done
python - <<PY
import json, glob
for f in sorted(glob.glob("$OUT/ab_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], round(d["value"], 2), "pairs/s", round(d["ms_per_step"], 3), "ms", d.get("secondary_640x480", {}).get("value"))
PY
