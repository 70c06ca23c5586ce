#!/bin/bash
# Round-3 GPU pass m: direct form of K10 -- bit-for-bit tests, micro-benchmark (tile height auto / 32 / 64), end-to-end A/B.
set -u
TAG=${1:-r03m}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_fusion.py -m gpu -x -q 2>&1 | tail -3
for bm in auto 32 64; do
  if [ $bm = auto ]; then unset S2M2_FUSION_DIRECT_BM; else export S2M2_FUSION_DIRECT_BM=$bm; fi
  timeout 300 python tools/fusionbench_direct.py > $OUT/fusionbench_direct_bm$bm.txt 2>&1; echo "fusionbench bm=$bm rc=$?"; grep -v amdgpu.ids $OUT/fusionbench_direct_bm$bm.txt
done
unset S2M2_FUSION_DIRECT_BM
for rep in 1 2; do
  for v in "S2M2_FUSION_DIRECT=0" "S2M2_FUSION_DIRECT=1" "S2M2_FUSION_DIRECT_MAX=200000"; do
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
