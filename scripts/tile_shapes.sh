#!/bin/bash
cd text-compression_amd
cp libtextcomp.so /tmp/libtextcomp_orig.so
for cfg in 512_8 256_8 256_16 1024_8 512_4; do
  cp libtextcomp_$cfg.so libtextcomp.so
  cd ..; echo "== $cfg"; for d in 0 12; do TC_RADIX_VARIANT=0 TC_DIAG=$d python scripts/sort_bench.py 1073741824 48 $((d==0)) 2>&1 | grep ms/pass; done; cd text-compression_amd
done
cp /tmp/libtextcomp_orig.so libtextcomp.so
