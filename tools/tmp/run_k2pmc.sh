set -u
R=$PWD; OUT=$R/gpurun_out/r06y; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/pmc_kernel.sh r06y sinkhorn tools/k2_only.py c3 c5 > /dev/null 2>&1
cd /tmp
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE -T -f csv -d $OUT/traffic_sinkhorn -o a -- python $R/tools/k2_only.py c3 c5 > $OUT/traffic_sinkhorn.log 2>&1
cd $R
python tools/pmc_kernel_summary.py sinkhorn $(find $OUT/trace_sinkhorn -name "*kernel_trace.csv") $(find $OUT/traffic_sinkhorn -name "*counter_collection.csv") > $OUT/pmc_sinkhorn_traffic.txt 2>&1
cat $OUT/pmc_sinkhorn.txt $OUT/pmc_sinkhorn_traffic.txt
python tools/k2_determinism.py > $OUT/k2_determinism.txt 2>&1; tail -5 $OUT/k2_determinism.txt
python tools/k2_only.py > $OUT/k2_final.txt 2>&1; S2M2_LIB_SUFFIX=_k2old python tools/k2_only.py >> $OUT/k2_final.txt 2>&1; grep -v amdgpu $OUT/k2_final.txt
python -m pytest tests/test_hip_dispinit.py tests/test_hip_utils.py -m gpu -q 2>&1 | tail -3
