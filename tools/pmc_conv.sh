#!/bin/bash
# PMC passes for the conv kernel (run on the GPU box): SQ issue/wait breakdown, MFMA busy, LDS conflicts, L2 hit rate.
OUT=$PWD/gpurun_out/${1:-pmc_conv}; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
for T in 1 2; do
  rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES -T -f csv -d $OUT/sq_t$T -o c -- python $R/tools/conv_only.py --tile $T > $OUT/sq_t$T.log 2>&1
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -T -f csv -d $OUT/lds_t$T -o c -- python $R/tools/conv_only.py --tile $T > $OUT/lds_t$T.log 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -T -f csv -d $OUT/tcc_t$T -o c -- python $R/tools/conv_only.py --tile $T > $OUT/tcc_t$T.log 2>&1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE -T -f csv -d $OUT/tcp_t$T -o c -- python $R/tools/conv_only.py --tile $T > $OUT/tcp_t$T.log 2>&1
done
cd $R; tail -3 $OUT/*.log | head -60
