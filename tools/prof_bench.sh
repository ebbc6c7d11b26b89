#!/bin/bash
# usage: tools/prof_bench.sh <tag> : rocprofv3 kernel-trace stats + HBM PMC passes of `python3 bench.py`
TAG=$1
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/$TAG
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fuse --no-pmc > $R/gpurun_out/$TAG/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_fetch -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fuse --no-pmc > $R/gpurun_out/$TAG/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/pmc_write -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fuse --no-pmc > $R/gpurun_out/$TAG/pmc_write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/$TAG/pmc_tcc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fuse --no-pmc > $R/gpurun_out/$TAG/pmc_tcc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/$TAG/pmc_sq -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-fuse --no-pmc > $R/gpurun_out/$TAG/pmc_sq.log 2>&1
grep -h '"metric"' $R/gpurun_out/$TAG/*.log | cut -c1-300
echo done
