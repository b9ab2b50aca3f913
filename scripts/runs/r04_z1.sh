#!/bin/bash
mkdir -p gpurun_out
TC_SA_TRACE=1 timeout -k 10 300 python scripts/classes_bench.py 1073741824 acgt_nrun > gpurun_out/r04z1_nrun.txt 2> gpurun_out/r04z1_nrun.err; cut -c1-400 gpurun_out/r04z1_nrun.txt; grep -E "round 0|ranks|round:" gpurun_out/r04z1_nrun.err | tail -n 90 | awk '{a[$0]++} END{for(k in a) print a[k], k}' | sort -k5 -n | head -5; grep -c "round: groups" gpurun_out/r04z1_nrun.err
