#!/bin/bash
# A/B the 1 GiB decode of prebuilt libtextcomp_<tag>.so variants (same box)
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so
for tag in "$@"; do cp libtextcomp_$tag.so libtextcomp.so; cd ..; echo "== $tag"; python scripts/decode_bench.py 2>/dev/null | tail -1; cd text-compression_amd; done
cp /tmp/libtextcomp_orig.so libtextcomp.so; cd ..; echo "== default"; python scripts/decode_bench.py 2>/dev/null | tail -1
