set -u
R=$PWD; OUT=$R/gpurun_out/r06y; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
timeout 240 rocprofv3 --pmc FETCH_SIZE -T -f csv -d $OUT/fetch_sinkhorn -o a -- python $R/tools/k2_only.py c3 c5 > $OUT/fetch_sinkhorn.log 2>&1; echo fetch rc=$?
timeout 240 rocprofv3 --pmc WRITE_SIZE -T -f csv -d $OUT/write_sinkhorn -o a -- python $R/tools/k2_only.py c3 c5 > $OUT/write_sinkhorn.log 2>&1; echo write rc=$?
cd $R
timeout 120 python tools/pmc_kernel_summary.py sinkhorn $(find $OUT/trace_sinkhorn -name "*kernel_trace.csv") $(find $OUT/fetch_sinkhorn $OUT/write_sinkhorn -name "*counter_collection.csv") > $OUT/pmc_sinkhorn_traffic.txt 2>&1
cat $OUT/pmc_sinkhorn_traffic.txt
timeout 300 python tools/k2_determinism.py > $OUT/k2_determinism.txt 2>&1; tail -5 $OUT/k2_determinism.txt
(timeout 200 python tools/k2_only.py; S2M2_LIB_SUFFIX=_k2old timeout 200 python tools/k2_only.py) 2>&1 | grep -v amdgpu > $OUT/k2_final.txt; cat $OUT/k2_final.txt
timeout 600 python -m pytest tests/test_hip_dispinit.py tests/test_hip_utils.py -m gpu -q 2>&1 | tail -3
