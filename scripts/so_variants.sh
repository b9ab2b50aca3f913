#!/bin/bash
# bench several prebuilt libtextcomp_<tag>.so variants
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so
for tag in "$@"; do cp libtextcomp_$tag.so libtextcomp.so; cd ..; echo "== $tag"; bash scripts/bench_brief.sh --steps 3 --warmup 1; cd text-compression_amd; done
cp /tmp/libtextcomp_orig.so libtextcomp.so; cd ..; echo "== default"; bash scripts/bench_brief.sh --steps 3 --warmup 1
