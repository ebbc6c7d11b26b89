#!/bin/bash
# usage: tools/prof_band.sh <tag> : SQ counters of the band worker vs the row worker on the bench workload (B = 32, 2 sweeps)
TAG=$1
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$TAG/pmc_a -- python3 $R/tools/sweep_knobs.py 32 2 :1 :1:0::1 > $R/gpurun_out/$TAG/pmc_a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/$TAG/pmc_b -- python3 $R/tools/sweep_knobs.py 32 2 :1 :1:0::1 > $R/gpurun_out/$TAG/pmc_b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("pmc_a", "pmc_b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$R/gpurun_out/$TAG/%s/*/*_counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "sweep_kernel" in r["Kernel_Name"] or "band_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k, {c: "%.4g" % (sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
