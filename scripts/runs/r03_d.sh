#!/bin/bash
# which workspace allocation recipe lands in the fast mode, and how often: 4 fresh processes per recipe
mkdir -p gpurun_out
out=gpurun_out/r03d_recipes.txt; : > $out
for rep in 1 2 3 4; do
  for k in 0 24 27 28 29 30 31 32 34; do
    r=$(TC_WS_VMM=$k timeout -k 10 120 python scripts/dbg/mode_place.py 1 2>&1 | grep PLACEMENTS_MS)
    echo "rep $rep vmm_chunk_log2 $k : $r" | tee -a $out
  done
done
