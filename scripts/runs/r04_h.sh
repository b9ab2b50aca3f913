#!/bin/bash
# round 4: the fine key directory (sparse rank lookups) and the far backward scan of the MTF list, against the oracle; then the classes they are for
set -o pipefail
out=gpurun_out/r04_h.txt; : > $out
run() { echo "== $*" | tee -a $out; ( "$@" 2>&1 | tail -n 2 ) | tee -a $out; }
TC_SA_ACCEL_MIN=1 TC_SA_SEG_MIN=1 run timeout -k 10 300 python tests/long/fuzz_long.py 300 71 120000 || exit 1
TC_SA_ACCEL_MIN=1 TC_SA_MSD=2 TC_SA_MSD_MIN_LOG2=10 run timeout -k 10 300 python tests/long/fuzz_long.py 200 72 300000 || exit 1
run timeout -k 10 300 python tests/long/fuzz_raw.py 600 73 || exit 1
grep -q "done:.* [1-9][0-9]* failures" $out && exit 1
python -m pytest tests/test_gpu_classes_digest.py tests/test_gpu_generators.py -q -x 2>&1 | tail -n 3 | tee -a $out
timeout -k 10 600 python scripts/classes_bench.py $((1<<30)) genome_like,binary2,runs_p0.9,repeat_4KiB 2>&1 | grep -v amdgpu.ids | cut -c1-210 | tee -a $out
