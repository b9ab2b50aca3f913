#!/bin/bash
# round 4 soak, part 5: the round's LAST build (chain rounds on dense and sparse ranks, hashed splitter offsets, the MTF
# retry skip), other seeds
mkdir -p gpurun_out
out=gpurun_out/r04_soak5.txt; : > $out
run() { echo "== $ENVS $*" | tee -a $out; ( "$@" 2>&1 | tail -n 1 ) | tee -a $out; }
ENVS="(defaults)"
run timeout -k 10 400 python tests/long/fuzz_long.py 1500 201 120000
run timeout -k 10 300 python tests/long/fuzz_raw.py 1200 202
run timeout -k 10 200 python tests/long/fuzz_fm.py 300 203
ENVS="TC_SA_CHAIN=2 TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1"; export TC_SA_CHAIN=2 TC_SA_SEG_MIN=1 TC_SA_ACCEL_MIN=1
run timeout -k 10 400 python tests/long/fuzz_chain.py 1500 204 60000
run timeout -k 10 400 python tests/long/fuzz_long.py 800 205 120000
ENVS="$ENVS TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10"; export TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10
run timeout -k 10 400 python tests/long/fuzz_long.py 600 206 300000
unset TC_SA_MSD TC_SA_MSD_MIN_LOG2 TC_SA_ACCEL_MIN
ENVS="TC_SA_CHAIN=2 TC_SA_SEG_MIN=1 TC_SA_DENSE=1"; export TC_SA_DENSE=1
run timeout -k 10 400 python tests/long/fuzz_chain.py 1000 207 120000
unset TC_SA_CHAIN TC_SA_SEG_MIN TC_SA_DENSE
ENVS="TC_IBWT_SEGCAP=64"; export TC_IBWT_SEGCAP=64
run timeout -k 10 300 python tests/long/fuzz_long.py 600 208 60000
unset TC_IBWT_SEGCAP
ENVS="TC_IBWT_LF=0 TC_DECODE_BYTES=0"; export TC_IBWT_LF=0 TC_DECODE_BYTES=0
run timeout -k 10 300 python tests/long/fuzz_long.py 500 209 60000
