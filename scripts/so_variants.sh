#!/bin/bash
# bench several prebuilt libtextcomp_<tag>.so variants
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so
for tag in "$@"; do cp libtextcomp_$tag.so libtextcomp.so; cd ..; echo "== $tag"; timeout -k 5 90 bash scripts/bench_brief.sh --steps 3 --warmup 1 || { cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so; echo "variant $tag failed or timed out: stopping"; exit 1; }; cd text-compression_amd; done
cp /tmp/libtextcomp_orig.so libtextcomp.so; cd ..; echo "== default"; timeout -k 5 90 bash scripts/bench_brief.sh --steps 3 --warmup 1
