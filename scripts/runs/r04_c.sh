#!/bin/bash
# round 4: generators + classes at a given size through bench-like calls; fuzz of the doubling rounds with the new skips
set -o pipefail
out=gpurun_out/r04_c.txt; : > $out
run() { echo "== $*" | tee -a $out; ( "$@" 2>&1 | tail -n 2 ) | tee -a $out; }
TC_SA_SEG_MIN=1 run timeout -k 10 300 python tests/long/fuzz_long.py 250 61 120000 || exit 1
TC_SA_SEG_MIN=1 TC_SA_DENSE=1 TC_SA_BIN_MIN_LOG2=0 run timeout -k 10 300 python tests/long/fuzz_long.py 200 62 200000 || exit 1
TC_SA_SEG_MIN=1 TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 run timeout -k 10 300 python tests/long/fuzz_long.py 150 63 400000 || exit 1
grep -q "done:.* [1-9][0-9]* failures" $out && exit 1
timeout -k 10 600 python - >> $out 2>&1 <<'PY'
import sys, os, json, ctypes as C
sys.path.insert(0, "text-compression_amd"); sys.path.insert(0, ".")
import torch, textcomp, bench
ctx = textcomp.Context(0)
for lg in (28, 30):
    r = bench.classes_leg(ctx, ctx.lib, torch, 1 << lg)
    print(json.dumps(r))
PY
