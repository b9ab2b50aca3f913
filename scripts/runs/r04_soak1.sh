#!/bin/bash
# round 4 soak, part 1: long randomized runs against the oracle on the round's last build
mkdir -p gpurun_out
out=gpurun_out/r04_soak1.txt; : > $out
run() { echo "== $*" | tee -a $out; ( "$@" 2>&1 | tail -n 2 ) | tee -a $out; }
run timeout -k 10 420 python tests/long/fuzz_long.py 2000 141 120000
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 run timeout -k 10 420 python tests/long/fuzz_long.py 1000 142 400000
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 TC_SA_MSD_BIG=1 TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1 run timeout -k 10 300 python tests/long/fuzz_long.py 500 143 300000
