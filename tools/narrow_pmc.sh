#!/bin/bash
# PMC passes over the K12 launches (run on the GPU box):   bash tools/narrow_pmc.sh <tag>   -> gpurun_out/<tag>/narrow_pmc.txt
OUT=$PWD/gpurun_out/${1:-narrow_pmc}; mkdir -p $OUT; R=$PWD; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_WAVES GRBM_GUI_ACTIVE -T -f csv -d $OUT/tcp -o n -- python $R/tools/narrow_only.py > $OUT/tcp.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS -T -f csv -d $OUT/sq -o n -- python $R/tools/narrow_only.py > $OUT/sq.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -T -f csv -d $OUT/fetch -o n -- python $R/tools/narrow_only.py > $OUT/fetch.log 2>&1
cd $R
python - > $OUT/narrow_pmc.txt <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "conv_narrow" in n or "conv_px" in n:
            acc[(n, int(r["Grid_Size"]), int(r["Workgroup_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key in sorted(acc):
    print("%s  grid %d threads, %d per block" % key)
    for c, v in sorted(acc[key].items()):
        print("    %-32s mean per launch %16.1f   (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
cat $OUT/narrow_pmc.txt
