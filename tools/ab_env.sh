#!/bin/bash
# Alternating same-box A/B of the bench under environment switches:   bash tools/ab_env.sh <out.txt> <reps> "VAR=a" "VAR=b" ...
# (each configuration: variables separated by ','; no CPU baseline, no secondary models)
OUT=$1; REPS=$2; shift 2
: > $OUT
for rep in $(seq 1 $REPS); do
  for cfg in "$@"; do
    env ${cfg//,/ } timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > /tmp/ab_tmp.json 2>> $OUT.err
    python - >> $OUT <<PY
import json
try:
    d = json.load(open("/tmp/ab_tmp.json"))
    print("rep $rep  %-40s ms_per_step %.4f  pairs/s %.2f  K1 %.2f us  launches %d" % ("$cfg", d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"], sum(d["forward"]["launches_by_family"].values())))
except Exception as e:
    print("rep $rep  $cfg  FAILED", e)
PY
  done
done
cat $OUT
