#!/bin/bash
# PMC passes (SQ set, LDS set; separate from the kernel trace) for the kernels one script launches:
#   bash tools/pmc_kernel.sh <tag> <kernel-name substring> <script.py> [args...]      -> gpurun_out/<tag>/pmc_<substring>.txt
TAG=$1; SUB=$2; shift 2
OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/trace_$SUB -o a -- python $R/"$@" > $OUT/trace_$SUB.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -T -f csv -d $OUT/sq_$SUB -o a -- python $R/"$@" > $OUT/sq_$SUB.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM -T -f csv -d $OUT/lds_$SUB -o a -- python $R/"$@" > $OUT/lds_$SUB.log 2>&1
cd $R
python tools/pmc_kernel_summary.py $SUB $(find $OUT/trace_$SUB -name "*kernel_trace.csv") $(find $OUT/sq_$SUB $OUT/lds_$SUB -name "*counter_collection.csv") > $OUT/pmc_$SUB.txt 2>&1
cat $OUT/pmc_$SUB.txt
