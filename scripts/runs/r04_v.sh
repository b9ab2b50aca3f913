#!/bin/bash
# experiment: chain rounds in EVERY dense round (TC_SA_CHAIN=2) on the classes that are not periodic -- do they save rounds?
mkdir -p gpurun_out
for e in 1 2; do
  echo "TC_SA_CHAIN=$e"
  TC_SA_CHAIN=$e timeout -k 10 400 python scripts/classes_bench.py 268435456 zipf_words,binary_words,runs_p0.9,binary2 2>/dev/null | cut -c1-330
done > gpurun_out/r04v_chain_all.txt 2>&1
cat gpurun_out/r04v_chain_all.txt
