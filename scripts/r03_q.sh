#!/bin/bash
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so; cp libtextcomp_prof.so libtextcomp.so; cd ..
TC_BENCH_PLACE=0 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-fm 2>&1 | grep -E "msd level|level [0-9]," | tail -n 12
cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so
