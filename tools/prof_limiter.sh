#!/bin/bash
# usage: tools/prof_limiter.sh <tag> [bench args] : the SQ counter passes behind DESIGN.md section 3.1 ("what limits the sweep
# kernel"), on `python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fuse` (one --pmc pass per counter set, no tracing
# combined with it).  Condensed by tools/summarize_limiter.py into profiles/<round>_limiter.json.
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-fuse $* --no-pmc"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$TAG/pmc_a -- python3 $R/bench.py $ARGS > $R/gpurun_out/$TAG/pmc_a.log 2>&1 &&
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/$TAG/pmc_b -- python3 $R/bench.py $ARGS > $R/gpurun_out/$TAG/pmc_b.log 2>&1 &&
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/$TAG/pmc_c -- python3 $R/bench.py $ARGS > $R/gpurun_out/$TAG/pmc_c.log 2>&1
grep -h '"metric"' $R/gpurun_out/$TAG/*.log | cut -c1-200
echo done
