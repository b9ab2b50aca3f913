#!/bin/bash
# long randomized runs against the oracle on the round's final build
mkdir -p gpurun_out
out=gpurun_out/r03_soak.txt; : > $out
run() { echo "== $*" | tee -a $out; ( "$@" 2>&1 | tail -n 3 ) | tee -a $out; }
run timeout -k 10 500 python tests/long/fuzz_long.py 2000 41 120000
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 run timeout -k 10 500 python tests/long/fuzz_long.py 1200 42 400000
TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 TC_SA_MSD_BIG=1 run timeout -k 10 400 python tests/long/fuzz_long.py 500 43 300000
TC_SA_BIN_MIN_LOG2=0 TC_MTF_TS=2 run timeout -k 10 400 python tests/long/fuzz_long.py 800 44 200000
run timeout -k 10 300 python tests/long/fuzz_raw.py 1500 45
run timeout -k 10 300 python tests/long/fuzz_fm.py 500 46
run timeout -k 10 400 python tests/long/boundary_sweep.py
