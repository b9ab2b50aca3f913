#!/bin/bash
# round 4 soak, part 4 (second session's build: chain rounds, bounded MTF far scan, one-atomic table build): long randomized
# runs against the oracle
mkdir -p gpurun_out
out=gpurun_out/r04_soak4.txt; : > $out
run() { echo "== $ENVS $*" | tee -a $out; ( "$@" 2>&1 | tail -n 1 ) | tee -a $out; }
ENVS="(defaults)"
run timeout -k 10 400 python tests/long/fuzz_long.py 1500 171 120000
run timeout -k 10 300 python tests/long/fuzz_chain.py 600 172 200000
run timeout -k 10 300 python tests/long/fuzz_raw.py 1500 173
run timeout -k 10 200 python tests/long/fuzz_fm.py 300 174
ENVS="TC_SA_CHAIN=2 TC_SA_DENSE=1 TC_SA_SEG_MIN=1"; export TC_SA_CHAIN=2 TC_SA_DENSE=1 TC_SA_SEG_MIN=1
run timeout -k 10 400 python tests/long/fuzz_chain.py 1500 175 60000
run timeout -k 10 400 python tests/long/fuzz_long.py 800 176 120000
ENVS="$ENVS TC_SA_BIN_MIN_LOG2=0"; export TC_SA_BIN_MIN_LOG2=0
run timeout -k 10 300 python tests/long/fuzz_chain.py 600 177 120000
unset TC_SA_CHAIN TC_SA_DENSE TC_SA_SEG_MIN TC_SA_BIN_MIN_LOG2
ENVS="TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1"; export TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1
run timeout -k 10 300 python tests/long/fuzz_long.py 800 178 150000
unset TC_SA_SEG_MIN TC_SA_ACCEL_MIN
ENVS="TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10"; export TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10
run timeout -k 10 400 python tests/long/fuzz_long.py 800 179 400000
unset TC_SA_MSD TC_SA_MSD_MIN_LOG2
run timeout -k 10 300 python tests/long/boundary_sweep.py
