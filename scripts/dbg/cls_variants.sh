#!/bin/bash
# one input class of classes_bench.py at 1 GiB with prebuilt library variants, on one box: $1 = class, rest = tags
CLS=$1; shift
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so; cd ..
for tag in "$@" default; do
  [ $tag = default ] && cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so || cp text-compression_amd/libtextcomp_$tag.so text-compression_amd/libtextcomp.so
  echo "== $tag"; timeout -k 5 150 python scripts/classes_bench.py 1073741824 $CLS 2>&1 | grep -v amdgpu | cut -c1-40,150-250 || { cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so; exit 1; }
done
cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so
