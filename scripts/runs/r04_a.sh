#!/bin/bash
# round 4, first look: per-step wall times and group-size classes of the doubling rounds (TC_SA_TRACE=1) per input class at 1 GiB
set -o pipefail
export TC_SA_TRACE=1
for c in genome_like zipf_words runs_p0.9 binary2 repeat_4KiB; do
  echo "=== $c" >> gpurun_out/r04_a_trace.txt
  timeout -k 10 300 python scripts/classes_bench.py $((1<<30)) $c >> gpurun_out/r04_a_trace.txt 2>&1 || exit 1
done
