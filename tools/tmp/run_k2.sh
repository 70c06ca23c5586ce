set -u
R=$PWD; OUT=$R/gpurun_out/r06x; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests/test_hip_utils.py tests/test_hip_dispinit.py -m gpu -q -x 2>&1 | tail -4 > $OUT/tests.txt; cat $OUT/tests.txt
(for i in 1 2; do S2M2_LIB_SUFFIX=_k2old python tools/k2_only.py c3 c2 c5 c3nopos; S2M2_K2_GL8=0 python tools/k2_only.py c3 c2; python tools/k2_only.py c3 c2 c5 c3nopos; done) 2>&1 | grep -v amdgpu > $OUT/k2_ab3.txt; cat $OUT/k2_ab3.txt
