#!/bin/bash
# usage: tools/prof_pmc2.sh <outdir> <NW>
OUT=$1; NW=$2
export TMPDIR=/tmp
export HCMVS_WAVES_PER_ROW=$NW
R=$PWD
mkdir -p $R/gpurun_out/$OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$OUT/pmc1 -- python3 $R/tools/quick_time.py 1920 1080 1600 8 8 > $R/gpurun_out/$OUT/pmc1.log 2>&1
echo done
