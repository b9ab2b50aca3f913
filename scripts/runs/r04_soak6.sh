#!/bin/bash
# round 4 soak, part 6: the final build at larger sizes -- round trips at 2^24 .. 2^28 +- 1, long fuzz cases up to 4 MB
mkdir -p gpurun_out
out=gpurun_out/r04_soak6.txt; : > $out
run() { echo "== $ENVS $*" | tee -a $out; ( "$@" 2>&1 | tail -n 1 ) | tee -a $out; }
ENVS="(defaults)"
run timeout -k 10 900 python tests/long/roundtrip_big.py
run timeout -k 10 500 python tests/long/fuzz_long.py 250 241 4000000
run timeout -k 10 500 python tests/long/fuzz_chain.py 250 242 4000000
ENVS="TC_SA_CHAIN=2 TC_SA_SEG_MIN=1"; export TC_SA_CHAIN=2 TC_SA_SEG_MIN=1
run timeout -k 10 500 python tests/long/fuzz_chain.py 200 243 2000000
