#!/bin/bash
# One pass on the GPU box (via gpurun):   bash tools/gpu_pass.sh <tag> [steps...]     steps: tests bench k1modes k1ab prof pmc parity configs
# Everything lands under gpurun_out/<tag>/ ; copy what is worth keeping into profiles/rNN/ (tools/collect_profiles.py).
set -u
TAG=${1:-pass}; shift
STEPS=${*:-tests bench}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $STEPS " == *" $1 "* ]]; }
python -c "import s2m2_amd.hip as h; h.load(); print('lib ok, ABI', h.ABI_VERSION)" > $OUT/load.log 2>&1 || { cat $OUT/load.log; python -m s2m2_amd.build > $OUT/build.log 2>&1; }
git -C $R rev-parse HEAD > $OUT/head.txt 2>/dev/null
if has tests; then
  timeout ${TEST_TIMEOUT:-1500} python -m pytest ${TEST_PATHS:-tests} -m gpu -q --maxfail=${MAXFAIL:-25} -rfs -p no:cacheprovider ${PYTEST_ARGS:-} > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
  tail -25 $OUT/pytest_gpu.txt
fi
if has smoke; then
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; echo "smoke rc=$?" >> $OUT/smoke.txt; tail -3 $OUT/smoke.txt
fi
if has bench; then
  timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_N1.json 2> $OUT/bench_N1.err; echo "bench rc=$?"
  python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_N1.json"))
    print("value", round(d["value"], 2), "ms", round(d["ms_per_step"], 3), "K1 us", round(d["roofline"]["avg_launch_us"], 2), "frac", round(d["roofline"]["frac"], 4),
          "attn", round(d["roofline_attention"]["frac"], 4), "fwd", round(d["forward"]["frac_of_mfma_peak"], 4), "launches", sum(d["forward"]["launches_by_family"].values()),
          "cpu", d.get("cpu_baseline", {}).get("kind"), d.get("cpu_baseline", {}).get("seconds_per_pair"))
except Exception as e:
    print("bench line unreadable:", e); print(open("$OUT/bench_N1.err").read()[-3000:])
PY
fi
if has k1modes; then
  : > $OUT/k1_store_modes.txt
  for CASE in ${K1MODES_CASES:-c3}; do timeout 600 python tools/k1_modes.py $CASE >> $OUT/k1_store_modes.txt 2>&1; done
  cat $OUT/k1_store_modes.txt
fi
if has k1ab; then
  # fold-vs-own-LayerNorm and store mode, alternating same-box runs of the bench (no CPU baseline, no secondary)
  : > $OUT/ab_k1.txt
  # K1AB_CFGS: configurations separated by ';', variables of one configuration by ','   (default: fold vs own LayerNorm)
  IFS=';' read -ra CFGS <<< "${K1AB_CFGS:-S2M2_FUSE_K1LN=1;S2M2_FUSE_K1LN=0}"
  for rep in $(seq 1 ${K1AB_REPS:-5}); do
    for cfg in "${CFGS[@]}"; do
      env ${cfg//,/ } timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/ab_tmp.json 2>> $OUT/ab_k1.err
      python - >> $OUT/ab_k1.txt <<PY
import json
try:
    d = json.load(open("$OUT/ab_tmp.json"))
    print("rep $rep  %-44s ms_per_step %.4f  pairs/s %.2f  K1 %.2f us  frac %.4f" % ("$cfg", d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
except Exception as e:
    print("rep $rep  $cfg  FAILED", e)
PY
    done
  done
  cat $OUT/ab_k1.txt
fi
if has micro; then
  timeout 600 python tools/pwbench.py > $OUT/pwbench.txt 2>&1; cat $OUT/pwbench.txt
  timeout 600 python tools/attnbench.py > $OUT/attnbench.txt 2>&1; cat $OUT/attnbench.txt
fi
if has parity; then
  timeout 1200 python tools/parity_report.py $OUT/parity_tables.txt c1 c1r3 c3r1 c2h noise_c1 noise_c3 > $OUT/parity_report.log 2>&1; tail -3 $OUT/parity_report.log
fi
if has configs; then
  timeout 900 python tools/configs_run.py > $OUT/configs_all_models.txt 2>&1; cat $OUT/configs_all_models.txt
fi
if has prof; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/prof_bench.log 2>&1
  echo "rocprof bench rc=$?"
  timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_k1_c3 -o k1 -- python $R/tools/k1_only.py --case c3 > $OUT/prof_k1_c3.log 2>&1
  cd $R
  timeout 600 python tools/layer_trace.py > $OUT/layer_trace_eager.txt 2>&1; tail -5 $OUT/layer_trace_eager.txt
  timeout 600 python tools/kbench.py --iters 30 > $OUT/kbench.txt 2>&1; tail -30 $OUT/kbench.txt
fi
if has pmc; then
  cd /tmp
  for CASE in c3; do
    timeout 300 rocprofv3 --pmc FETCH_SIZE -T -f csv -d $OUT/pmc_fetch_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/pmc_fetch_$CASE.log 2>&1
    timeout 300 rocprofv3 --pmc WRITE_SIZE -T -f csv -d $OUT/pmc_write_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/pmc_write_$CASE.log 2>&1
  done
  cd $R
fi
find $OUT -name "*_kernel_stats.csv" | head; du -sh $OUT
