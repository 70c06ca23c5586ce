#!/bin/bash
# Run on the GPU box (via gpurun): parity tests, bench line, rocprofv3 kernel stats of the bench, PMC passes for K1.
# Everything lands under gpurun_out/<tag>/ ; copy the summaries worth keeping into profiles/.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m s2m2_amd.build > $OUT/build.log 2>&1
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_gpu.log
  tail -5 $OUT/pytest_gpu.log
fi
timeout 900 python bench.py --steps ${STEPS:-10} --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cat $OUT/bench.json
timeout 600 python tools/kbench.py --iters 30 > $OUT/kbench.log 2>&1; cp gpurun_out/kbench.json $OUT/ 2>/dev/null
cat $OUT/kbench.log
R=$PWD
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
echo "rocprof bench rc=$?"
for CASE in c3 c2; do
  timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_k1_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/prof_k1_$CASE.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -T -f csv -d $OUT/pmc_fetch_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/pmc_fetch_$CASE.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -T -f csv -d $OUT/pmc_write_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/pmc_write_$CASE.log 2>&1
done
cd $R
find $OUT -name "*.csv" | head -40
du -sh $OUT
