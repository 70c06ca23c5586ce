#!/bin/bash
# Bench line + rocprofv3 kernel stats of the same command only (box-to-box spread check).   usage: bash tools/gpu_bench_only.sh <tag>
set -u
TAG=${1:-r02b}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_N1.json 2> $OUT/bench_N1.err; echo "bench rc=$?"
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
echo "rocprof bench rc=$?"
cd $R
python -c "import json; d=json.load(open('$OUT/bench_N1.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_launch_us'])"
