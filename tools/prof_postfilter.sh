#!/bin/bash
# usage: tools/prof_postfilter.sh <tag> [n_images=16] : per-kernel time of the authors' schedule (4 outer x 3 inner sweeps, 10 views, post-filters
# after outer iterations 1 and 2) on an N x 1080p ring scene through the stand-alone driver -- which kernels the post-filter's fusions spend
# their time in.  rocprofv3 --kernel-trace --stats on the driver binary itself.
TAG=$1; N=${2:-16}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$TAG
SCENE=$(python3 tools/authors_schedule.py $N 1920 1080 --scene-only | tail -1)
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- $R/hc-mvs_amd/DensifyPointCloud --input-file $SCENE/scene.mvs -w $SCENE -o $SCENE/scene_dense.mvs \
  --verbosity 2 --fusion-mode 0 --min-resolution 100 --number-views 10 --resolution-level 0 --n-EstimationIters 3 --n-EstimationIters-external 4 \
  --n-photometric_flow 0.26 --n-nOptimize 1 --n-adapthalfwin 7 --n-propagatehalfwin 5 --n-propagatestep 4 --resume 0 > $R/gpurun_out/$TAG/run.log 2>&1
cat $R/gpurun_out/$TAG/run.log | grep -v "^[EWI]2026" | tail -8
F=$(find $R/gpurun_out/$TAG/trace -name "*kernel_stats.csv" | head -1)
cp $F $R/gpurun_out/$TAG/kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-70s calls %6s total %9.2f ms avg %9.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, 100*float(r["TotalDurationNs"])/tot))
PY
