#!/bin/bash
cd text-compression_amd
cp libtextcomp.so /tmp/libtextcomp_orig.so
for cfg in "$@"; do
  cp libtextcomp_$cfg.so libtextcomp.so
  cd ..; echo "== $cfg"; python scripts/sort_bench.py 1073741824 32 1 2>&1 | grep ms/pass; bash scripts/bench_brief.sh --steps 2 --warmup 1; cd text-compression_amd
done
cp /tmp/libtextcomp_orig.so libtextcomp.so
