#!/bin/bash
# diagnostic: the sweep worker's tickets as whole rows (0) or stretches of N columns, small batches of 1080p images,
# one launch per sweep or one launch for all sweeps
out=gpurun_out/${1:-r04seg}; mkdir -p $out
for b in ${2:-6 8 12}; do
  for mode in per-sweep one; do
    for seg in 0 256; do
      echo "== batch $b, HCMVS_SWEEP_LAUNCHES=$mode HCMVS_SWEEP_SEGMENT=$seg" >> $out/knobs.txt
      HCMVS_SWEEP_LAUNCHES=$mode HCMVS_SWEEP_SEGMENT=$seg timeout -k 10 300 python tools/sweep_knobs.py $b 8 ":1" >> $out/knobs.txt 2>&1 || exit 1
    done
  done
done
