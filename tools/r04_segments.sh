#!/bin/bash
# diagnostic: the sweep worker's tickets as whole rows (0) or stretches of N columns, small batches of 1080p images
out=gpurun_out/r04seg; mkdir -p $out
for b in 3 4 6 8 12; do
  for seg in 0 256 512; do
    echo "== batch $b, HCMVS_SWEEP_SEGMENT=$seg" >> $out/knobs.txt
    HCMVS_SWEEP_SEGMENT=$seg timeout -k 10 300 python tools/sweep_knobs.py $b 8 ":1" >> $out/knobs.txt 2>&1 || exit 1
  done
done
