#!/bin/bash
# round 4 soak, part 3: other seeds, the selectors the first two parts did not combine
mkdir -p gpurun_out
out=gpurun_out/r04_soak3.txt; : > $out
run() { echo "== $ENVS $*" | tee -a $out; ( "$@" 2>&1 | tail -n 1 ) | tee -a $out; }
ENVS="TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 TC_SA_MSD_KEYONLY=2 TC_SA_SEG_MIN=1"; export TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 TC_SA_MSD_KEYONLY=2 TC_SA_SEG_MIN=1
run timeout -k 10 400 python tests/long/fuzz_long.py 1000 161 400000
unset TC_SA_MSD TC_SA_MSD_MIN_LOG2 TC_SA_MSD_KEYONLY TC_SA_SEG_MIN
ENVS="TC_SA_SEG_MIN=1 TC_SA_TINY=0 TC_SA_KDIR_SEARCH=1 TC_SA_ACCEL_MIN=1"; export TC_SA_SEG_MIN=1 TC_SA_TINY=0 TC_SA_KDIR_SEARCH=1 TC_SA_ACCEL_MIN=1
run timeout -k 10 300 python tests/long/fuzz_long.py 800 162 150000
unset TC_SA_SEG_MIN TC_SA_TINY TC_SA_KDIR_SEARCH TC_SA_ACCEL_MIN
ENVS="TC_SA_SEG=0"; export TC_SA_SEG=0
run timeout -k 10 300 python tests/long/fuzz_long.py 600 163 150000
unset TC_SA_SEG
ENVS="TC_SA_SEG_MIN=1 TC_SA_SAMPLE=0 TC_SA_FINISH=0"; export TC_SA_SEG_MIN=1 TC_SA_SAMPLE=0 TC_SA_FINISH=0
run timeout -k 10 300 python tests/long/fuzz_long.py 600 164 150000
unset TC_SA_SEG_MIN TC_SA_SAMPLE TC_SA_FINISH
ENVS="TC_HOST_STAGED=0"; export TC_HOST_STAGED=0
run timeout -k 10 200 python tests/long/fuzz_long.py 300 165 120000
