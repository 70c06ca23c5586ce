#!/bin/bash
# PMC passes (counters only, separate runs) for the kernels DESIGN.md calls VALU-bound: K9 chain (weights-stationary and streaming form), K10, K2.
OUT=$PWD/gpurun_out/${1:-pmc_valu}; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp; cd /tmp
for V in ws stream; do
  E=""; [ $V = stream ] && E="S2M2_CHAIN_WS=0"
  env $E rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU -T -f csv -d $OUT/sq_$V -o v -- python $R/tools/valu_only.py > $OUT/sq_$V.log 2>&1
  env $E rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA -T -f csv -d $OUT/lds_$V -o v -- python $R/tools/valu_only.py > $OUT/lds_$V.log 2>&1
done
cd $R
python tools/pmc_valu_summary.py $OUT > $OUT/pmc_valu_summary.txt 2>&1; cat $OUT/pmc_valu_summary.txt
