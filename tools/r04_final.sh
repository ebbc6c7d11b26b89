#!/bin/bash
# the measurements committed under profiles/r04_* (one gpurun call): kernel trace + PMC passes of bench.py, the authors' schedule through
# the driver (incremental post-filter chain), its per-kernel profile, the 2-rank rehearsal of bench.py, and bench.py itself
R=$PWD
mkdir -p gpurun_out/r04final
tools/prof_bench.sh r04final/prof > gpurun_out/r04final/prof_bench.log 2>&1
python3 tools/authors_schedule.py 64 > gpurun_out/r04final/authors_schedule_64x1080p.txt 2>&1
tools/prof_postfilter.sh r04final/pf 16 > gpurun_out/r04final/postfilter_profile.txt 2>&1
HCMVS_BENCH_ONE_DEVICE=1 HCMVS_BENCH_BACKEND=gloo python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 0 --batch 4 --no-pmc > gpurun_out/r04final/bench_2rank_rehearsal.json 2> gpurun_out/r04final/bench_2rank_rehearsal.err
python3 bench.py > gpurun_out/r04final/bench.json 2> gpurun_out/r04final/bench.err
tail -c 2500 gpurun_out/r04final/bench.json
echo; tail -c 600 gpurun_out/r04final/bench_2rank_rehearsal.json
