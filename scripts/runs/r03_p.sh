#!/bin/bash
# A/B of prebuilt library variants on ONE box: msd tests (correctness) + per-kernel times
mkdir -p gpurun_out
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so; cd ..
for tag in "$@" default; do
  [ $tag = default ] && cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so || cp text-compression_amd/libtextcomp_$tag.so text-compression_amd/libtextcomp.so
  echo "== $tag"
  timeout -k 10 300 python -m pytest tests/test_gpu_msd.py tests/test_gpu_encode.py -x -q 2>&1 | tail -n 2
  timeout -k 10 200 bash scripts/prof_brief.sh v_$tag --no-fm 2>&1 | head -n 7
  grep '^{' gpurun_out/prof_v_${tag}_bench.log | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   step %.2f ms, container step %.2f' % (d['ms_per_step'], d['container']['ms_per_step_with_container']))"
done
cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so
