#!/bin/bash
# Round-2 measurement pass on the GPU box (via gpurun): everything lands under gpurun_out/<tag>/; tools/collect_profiles.py then
# copies the summaries worth keeping into profiles/<tag>/.   usage: bash tools/gpu_profile_r02.sh r02
set -u
TAG=${1:-r02}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# 1. the bench line exactly as the driver runs it
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_N1.json 2> $OUT/bench_N1.err; echo "bench rc=$?"
S2M2_CV_BAND=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_N1_banded.json 2>/dev/null
# 2. rocprofv3 kernel stats of the same command (no CPU baseline: that leg is host-only)
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/prof_bench.log 2>&1
echo "rocprof bench rc=$?"
# 3. K1 alone: stats + PMC passes (separate runs per counter)
for CASE in c3 c2; do
  timeout 300 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_k1_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/prof_k1_$CASE.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -T -f csv -d $OUT/pmc_fetch_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/pmc_fetch_$CASE.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -T -f csv -d $OUT/pmc_write_$CASE -o k1 -- python $R/tools/k1_only.py --case $CASE > $OUT/pmc_write_$CASE.log 2>&1
done
cd $R
# 4. micro-benchmarks and timelines
timeout 600 python tools/kbench.py --iters 30 2>&1 | grep -v amdgpu.ids > $OUT/kbench.txt
timeout 600 python tools/convbench.py --cold --only 3x --tiles 26,24 2>&1 | grep -v amdgpu.ids > $OUT/convbench_3x3_cold.txt
timeout 300 python tools/attnbench.py 2>&1 | grep -v amdgpu.ids > $OUT/attnbench.txt
timeout 200 python tools/fusionbench.py 2>&1 | grep -v amdgpu.ids > $OUT/fusionbench.txt
S2M2_LIB_SUFFIX=_k1trace timeout 120 python tools/k1_trace.py c3 2>&1 | grep -v amdgpu.ids > $OUT/k1_timeline.txt
S2M2_LIB_SUFFIX=_fragtrace timeout 120 python tools/frag_trace.py 2>&1 | grep -v amdgpu.ids > $OUT/frag_timeline.txt
timeout 300 python tools/layer_trace.py --ab S2M2_CONV_FRAG=0,1 --iters 5 2>&1 | grep -v amdgpu.ids > $OUT/layer_ab_frag.txt
timeout 300 python tools/layer_trace.py --iters 3 2>&1 | grep -v amdgpu.ids > $OUT/layer_trace_eager.txt
# 5. parity tables at the BASELINE sizes + the other configurations through the drop-in module
timeout 600 python tools/parity_report.py $OUT/parity_c1_c3_c2.txt > /dev/null 2>&1
timeout 900 python tools/configs_run.py 2>&1 | grep -v amdgpu.ids > $OUT/configs_all_models.txt
du -sh $OUT
