#!/bin/bash
# timing-only ablations of the radix pass (libtextcomp_diag.so = build with -DTC_RADIX_DIAG; wrong output)
cp text-compression_amd/libtextcomp.so /tmp/libtextcomp_orig.so
cp text-compression_amd/libtextcomp_diag.so text-compression_amd/libtextcomp.so
for d in 0 128 384 126 254 510 2 4 8 16; do
  printf "TC_DIAG=%-4s " $d; TC_DIAG=$d python scripts/sort_bench.py 1073741824 32 0 2>&1 | grep -o "[0-9.]* ms/pass"
done
cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so
