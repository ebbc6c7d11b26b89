#!/bin/bash
# usage: tools/prof_pmc.sh <outdir> <NW> ; collects kernel trace + two PMC passes for the 1080p/8-view/8-sweep estimate
OUT=$1; NW=$2
export TMPDIR=/tmp
export HCMVS_WAVES_PER_ROW=$NW
R=$PWD
mkdir -p $R/gpurun_out/$OUT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$OUT/trace -- python3 $R/tools/quick_time.py 1920 1080 1600 8 8 > $R/gpurun_out/$OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$OUT/pmc1 -- python3 $R/tools/quick_time.py 1920 1080 1600 8 8 > $R/gpurun_out/$OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/$OUT/pmc2 -- python3 $R/tools/quick_time.py 1920 1080 1600 8 8 > $R/gpurun_out/$OUT/pmc2.log 2>&1
echo done
