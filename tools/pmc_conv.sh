#!/bin/bash
# PMC passes for the conv kernel (run on the GPU box): SQ issue/wait breakdown, MFMA busy, LDS, VMEM.  usage: pmc_conv.sh tag "tiles" shape
OUT=$PWD/gpurun_out/${1:-pmc_conv}; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp; cd /tmp
for T in ${2:-5 8}; do
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -T -f csv -d $OUT/sq_t$T -o c -- python $R/tools/conv_only.py --tile $T --shape ${3:-11} > $OUT/sq_t$T.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM -T -f csv -d $OUT/lds_t$T -o c -- python $R/tools/conv_only.py --tile $T --shape ${3:-11} > $OUT/lds_t$T.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS -T -f csv -d $OUT/ins_t$T -o c -- python $R/tools/conv_only.py --tile $T --shape ${3:-11} > $OUT/ins_t$T.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -T -f csv -d $OUT/tcp_t$T -o c -- python $R/tools/conv_only.py --tile $T --shape ${3:-11} > $OUT/tcp_t$T.log 2>&1
done
cd $R
