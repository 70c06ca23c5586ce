#!/bin/bash
# rocprofv3 kernel stats of the L (c4), M and XL (c5) forwards through the drop-in module:   bash tools/gpu_models.sh <tag>
#   -> gpurun_out/<tag>/models/{c4,M,c5}_kernel_stats.csv + README_kernel_shares.txt (top kernels by share)
TAG=${1:-models}; R=$PWD; OUT=$R/gpurun_out/$TAG/models; mkdir -p $OUT; export TMPDIR=/tmp
echo "# rocprofv3 --kernel-trace --stats of tools/configs_run.py <config> (3 warm-up + 5 timed forwards through the drop-in module, fp16, plus the engine's own warm-up / capture forwards: the ms column divides the total by 8 and is an upper bound per forward; the percentages are exact)" > $OUT/README_kernel_shares.txt
for CFG in c4 M c5; do
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_$CFG -o m -- python $R/tools/configs_run.py $CFG > $OUT/$CFG.log 2>&1
  cd $R
  ST=$(find $OUT/prof_$CFG -name "*kernel_stats.csv" | head -1)
  [ -n "$ST" ] && cp $ST $OUT/${CFG}_kernel_stats.csv
  python - >> $OUT/README_kernel_shares.txt <<PY
import csv, re
line = [l for l in open("$OUT/$CFG.log") if l.startswith("$CFG:")]
print("=== $CFG:", line[0].strip()[:140] if line else "(no result line)")
try:
    rows = list(csv.DictReader(open("$OUT/${CFG}_kernel_stats.csv")))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    def short(n):
        n = re.sub(r"^_ZN4s2m2\d+", "", n)
        m = re.match(r"([a-z_0-9]+_kernel)I(.*?)E+v", n)
        if m:
            a = re.findall(r"Li(\d+)|Lb(\d)", m.group(2))
            return m.group(1).replace("_kernel", "") + "<" + ",".join(x or y for x, y in a) + ">"
        return re.sub(r"\(.*", "", n)[:60]
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:18]:
        t = float(r["TotalDurationNs"])
        print(f"{int(r['Calls']):5d} calls {t / 8e6:9.3f} ms {100 * t / tot:5.1f}%  avg {float(r['AverageNs']) / 1e3:9.1f} us  {short(r['Name'])}")
except Exception as e:
    print("  (no stats:", e, ")")
PY
  rm -rf $OUT/prof_$CFG
done
cat $OUT/README_kernel_shares.txt | head -70
