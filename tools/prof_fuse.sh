#!/bin/bash
# usage: tools/prof_fuse.sh <tag> : rocprofv3 kernel-trace stats of the fusion leg (tools/fuse_profile.py)
TAG=$1
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp
python3 $R/tools/fuse_profile.py > $R/gpurun_out/$TAG/plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/tools/fuse_profile.py > $R/gpurun_out/$TAG/trace.log 2>&1
cat $R/gpurun_out/$TAG/plain.log | tail -2
find $R/gpurun_out/$TAG/trace -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/$TAG/kernel_stats.csv \;
find $R/gpurun_out/$TAG/trace -name "*kernel_trace.csv" -exec cp {} $R/gpurun_out/$TAG/kernel_trace.csv \;
cut -c1-150 $R/gpurun_out/$TAG/kernel_stats.csv | head -30
rm -rf $R/gpurun_out/$TAG/trace
echo done
