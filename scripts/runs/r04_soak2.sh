#!/bin/bash
# round 4 soak, part 2
mkdir -p gpurun_out
out=gpurun_out/r04_soak2.txt; : > $out
run() { echo "== $*" | tee -a $out; ( "$@" 2>&1 | tail -n 2 ) | tee -a $out; }
TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1 run timeout -k 10 300 python tests/long/fuzz_long.py 800 144 200000
TC_SA_SEG_MIN=1 TC_SA_DENSE=1 TC_SA_BIN_MIN_LOG2=0 TC_MTF_TS=2 run timeout -k 10 300 python tests/long/fuzz_long.py 800 145 200000
run timeout -k 10 200 python tests/long/fuzz_raw.py 1500 146
run timeout -k 10 200 python tests/long/fuzz_fm.py 500 147
run timeout -k 10 300 python tests/long/boundary_sweep.py
