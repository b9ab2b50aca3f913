#!/bin/bash
# fused MTF+RLE kernel: do staggered starts of a CU's blocks overlap the store phase with the compute phases?
mkdir -p gpurun_out
for d in 0 1281 1282 1283 1284 2561 641; do
  TC_MTFRLE_STAGGER=$d bash scripts/prof_brief.sh r03u$d --no-fm 2>&1 | grep -E "mtf_rle" | sed "s/^/stagger=$d /"
done
