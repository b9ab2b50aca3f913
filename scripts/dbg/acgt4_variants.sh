#!/bin/bash
# iid ACGT at 1 GiB (the finish instance for buckets of ~4096 pairs) with prebuilt variants, on one box
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so; cd ..
for tag in "$@" default; do
  [ $tag = default ] && cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so || cp text-compression_amd/libtextcomp_$tag.so text-compression_amd/libtextcomp.so
  echo "== $tag"; timeout -k 5 120 python scripts/classes_bench.py 1073741824 acgt4 2>&1 | grep -v amdgpu | cut -c1-200 || { cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so; exit 1; }
done
cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so
