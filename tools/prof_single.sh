#!/bin/bash
# usage: tools/prof_single.sh <tag> : instruction-cache / scalar-cache / wait counters of the one-image sweep (sweep_knobs.py 1 8)
TAG=$1
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/$TAG/p$i -- python3 $R/tools/sweep_knobs.py ${2:-1} 8 ":1" > $R/gpurun_out/$TAG/p$i.log 2>&1 || echo "set $i failed"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$R/gpurun_out/$TAG/p*/**/*_counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "sweep_kernel" in row["Kernel_Name"]:
            acc[row["Kernel_Name"][:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()): print("   %-28s %.4g per launch (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
rm -rf $R/gpurun_out/$TAG/p[0-9]
echo done
