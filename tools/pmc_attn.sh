#!/bin/bash
# MFMA utilisation of the attention kernel (north_star: "MFMA utilisation on the attention GEMMs"): SQ_VALU_MFMA_BUSY_CYCLES over
# (SIMDs x kernel cycles), separate PMC passes, shapes of tools/attnbench.py.   usage: bash tools/pmc_attn.sh <tag>
OUT=$PWD/gpurun_out/${1:-pmc_attn}; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -T -f csv -d $OUT/trace -o a -- python $R/tools/attnbench.py > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -T -f csv -d $OUT/sq -o a -- python $R/tools/attnbench.py > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -T -f csv -d $OUT/lds -o a -- python $R/tools/attnbench.py > $OUT/lds.log 2>&1
cd $R; grep nb= $OUT/trace.log
