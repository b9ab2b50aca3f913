#!/bin/bash
# round 4 soak, part 7: the round's final build (ranks in the first group pass after a drowned finish pass)
mkdir -p gpurun_out
out=gpurun_out/r04_soak7.txt; : > $out
run() { echo "== $ENVS $*" | tee -a $out; ( "$@" 2>&1 | tail -n 1 ) | tee -a $out; }
ENVS="(defaults)"
run timeout -k 10 400 python tests/long/fuzz_long.py 1500 261 120000
run timeout -k 10 300 python tests/long/fuzz_chain.py 400 262 1000000
ENVS="TC_SA_SAMPLE=0"; export TC_SA_SAMPLE=0
run timeout -k 10 400 python tests/long/fuzz_long.py 1000 263 120000
ENVS="TC_SA_SAMPLE=0 TC_SA_CHAIN=2 TC_SA_SEG_MIN=1"; export TC_SA_CHAIN=2 TC_SA_SEG_MIN=1
run timeout -k 10 400 python tests/long/fuzz_chain.py 800 264 60000
