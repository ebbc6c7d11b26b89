#!/bin/bash
# usage: tools/prof_pmc_batch.sh <outdir> : SQ counter passes for one batched estimate (8 x 1080p / 8 views / 8 sweeps)
OUT=$1
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$OUT
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$OUT/pmc1 -- python3 $R/tools/quick_batch.py 16 1 > $R/gpurun_out/$OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/$OUT/pmc2 -- python3 $R/tools/quick_batch.py 16 1 > $R/gpurun_out/$OUT/pmc2.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc1", "pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$R/gpurun_out/$OUT/%s/*/*_counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "sweep_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
echo done
