set -u
R=$PWD; OUT=$R/gpurun_out/r06z2; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_dropin.py tests/test_hip_e2e.py -m gpu -q -x 2>&1 | tail -4
K1AB_REPS=4 K1AB_CFGS="S2M2_EAGER_PREP=0;S2M2_EAGER_PREP=1" bash tools/gpu_pass.sh r06z2 k1ab > /dev/null 2>&1
cat $OUT/ab_k1.txt
