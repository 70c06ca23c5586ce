#!/bin/bash
# Round-3 GPU pass p2: every model size (S / M / L / XL, fp16) through the drop-in module with the direct K9 / K10 forms on, and the fp16
# forward tests that compare against the fp32 path.
set -u
TAG=${1:-r03p2}
R=$PWD
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/configs_run.py 2>&1 | grep -v amdgpu.ids > $OUT/configs_all_models.txt; cat $OUT/configs_all_models.txt
timeout 300 python -m pytest tests/test_hip_dropin.py -m gpu -x -q 2>&1 | tail -2
