#!/bin/bash
# round 4: the segmented sort of the doubling rounds -- forced at every size against the oracle (sparse and dense ranks), then classes at 1 GiB
# usage: r04_b.sh [classes (comma separated)] [fuzz cases scale]
set -o pipefail
out=gpurun_out/r04_b.txt; : > $out
CLS=${1:-genome_like,zipf_words,runs_p0.9}
run() { echo "== $*" | tee -a $out; ( "$@" 2>&1 | tail -n 2 ) | tee -a $out; }
TC_SA_SEG_MIN=1 run timeout -k 10 300 python tests/long/fuzz_long.py 300 51 120000 || exit 1
TC_SA_SEG_MIN=1 TC_SA_DENSE=1 TC_SA_BIN_MIN_LOG2=0 run timeout -k 10 300 python tests/long/fuzz_long.py 200 52 200000 || exit 1
TC_SA_SEG_MIN=1 TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 run timeout -k 10 300 python tests/long/fuzz_long.py 200 53 400000 || exit 1
grep -q "0 failures" $out || exit 1
for c in ${CLS//,/ }; do
  TC_SA_TRACE=1 timeout -k 10 300 python scripts/classes_bench.py $((1<<30)) $c 2>&1 | grep -v "members by\|amdgpu.ids" | tail -n 22 >> $out || exit 1
done
