#!/bin/bash
# per-kernel averages of several prebuilt libtextcomp_<tag>.so variants on ONE box (boxes differ by ~10 %)
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so; cd ..
for tag in "$@" default; do
  [ $tag = default ] && cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so || cp text-compression_amd/libtextcomp_$tag.so text-compression_amd/libtextcomp.so
  echo "== $tag"; timeout -k 5 120 bash scripts/prof_brief.sh v_$tag | head -${TOPK:-5} || { cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so; exit 1; }
  tail -1 gpurun_out/prof_v_${tag}_bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('   step %.2f ms'%d['ms_per_step'])" 2>/dev/null
done
cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so
