#!/bin/bash
# round 4: seg_small with the tiny members through the network when that is free -- parity, then three classes at 1 GiB
set -o pipefail
out=gpurun_out/r04_p.txt; : > $out
TC_SA_SEG_MIN=1 timeout -k 10 300 python tests/long/fuzz_long.py 300 101 120000 2>&1 | tail -n 1 | tee -a $out
TC_SA_SEG_MIN=1 TC_SA_DENSE=1 TC_SA_BIN_MIN_LOG2=0 timeout -k 10 300 python tests/long/fuzz_long.py 200 102 200000 2>&1 | tail -n 1 | tee -a $out
TC_SA_SEG_MIN=1 TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 timeout -k 10 300 python tests/long/fuzz_long.py 150 103 400000 2>&1 | tail -n 1 | tee -a $out
timeout -k 10 300 python scripts/classes_bench.py 1073741824 zipf_words,repeat_4KiB 2>&1 | grep -v amdgpu | cut -c1-120 | tee -a $out
timeout -k 10 300 python scripts/classes_bench.py 1073741824 runs_p0.9,genome_like,binary_words 2>&1 | grep -v amdgpu | cut -c1-120 | tee -a $out
