#!/bin/bash
# Kernel trace + one SQ counter pass over the (eagerly launched) forward of the bench, then tools/pmc_forward.py:   bash tools/pmc_forward.sh <tag>
# -> gpurun_out/<tag>/pmc_forward.txt   (every rocprofv3 call under `timeout`: a counter set the hardware cannot collect aborts and then hangs in finalisation)
set -u
TAG=${1:-pmcfwd}; R=$PWD; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
cd /tmp
S2M2_GRAPH=0 timeout 500 rocprofv3 --kernel-trace -T -f csv -d $OUT/trace -o a -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/trace.log 2>&1; echo trace rc=$?
S2M2_GRAPH=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -T -f csv -d $OUT/sq -o a -- python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/sq.log 2>&1; echo pmc rc=$?
cd $R
timeout 120 python tools/pmc_forward.py $(find $OUT/trace -name "*kernel_trace.csv") $(find $OUT/sq -name "*counter_collection.csv") > $OUT/pmc_forward.txt 2>&1
cut -c1-170 $OUT/pmc_forward.txt
