#!/bin/bash
# same-box A/B over prebuilt libtextcomp_<tag>.so variants: headline step + one class; $1 = "class n", rest = tags
SPEC=$1; shift
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so
for tag in "$@" default; do
  if [ "$tag" = default ]; then cp /tmp/libtextcomp_orig.so libtextcomp.so; else cp libtextcomp_$tag.so libtextcomp.so; fi
  cd ..; echo "== $tag"; bash scripts/bench_brief.sh --steps 3 --warmup 1
  set -- $SPEC_DUMMY; python scripts/classes_bench.py ${SPEC#* } ${SPEC%% *} 2>/dev/null | cut -c1-150; cd text-compression_amd
done
cp /tmp/libtextcomp_orig.so libtextcomp.so
