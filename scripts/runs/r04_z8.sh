#!/bin/bash
mkdir -p gpurun_out
for e in 1 3; do echo "TC_SA_MSD=$e"; TC_SA_MSD=$e TC_SA_TRACE=0 timeout -k 10 300 python scripts/classes_bench.py 1073741824 acgt_nrun,genome_like 2>/dev/null | cut -c1-330; done
