set -u
R=$PWD; OUT=$R/gpurun_out/r06w; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_hip_utils.py tests/test_hip_norm.py tests/test_hip_dispinit.py -m gpu -q -x 2>&1 | tail -4 > $OUT/tests.txt; cat $OUT/tests.txt
K1AB_REPS=3 K1AB_CFGS="S2M2_LIB_SUFFIX=_k2old;S2M2_LIB_SUFFIX=" bash tools/gpu_pass.sh r06w k1ab > /dev/null 2>&1
cat $OUT/ab_k1.txt
cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/prof_bench -o bench -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/prof_bench.log 2>&1; cd $R
f=$(find $OUT/prof_bench -name "*kernel_stats.csv" | head -1); grep -i "stem_mlp\|groupnorm\|sinkhorn\|image_prep" $f | cut -c1-200
for n in 3 4; do S2M2_PAIR_STREAMS=$n python tools/batch_modes.py --child 2>&1 | grep -v amdgpu; done > $OUT/pair_streams_n.txt 2>&1; cat $OUT/pair_streams_n.txt
