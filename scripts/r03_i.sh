#!/bin/bash
mkdir -p gpurun_out
cd text-compression_amd; cp libtextcomp.so /tmp/libtextcomp_orig.so; cd ..
for tag in default nt256 default; do
  [ $tag = default ] && cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so || cp text-compression_amd/libtextcomp_$tag.so text-compression_amd/libtextcomp.so
  if [ $tag != default ] || [ -z "$done_default_tests" ]; then
    timeout -k 10 600 python -m pytest tests/test_gpu_container_fused.py tests/test_gpu_container.py "tests/test_gpu_encode.py" -x -q 2>&1 | tail -n 2
    done_default_tests=1
  fi
  TC_BENCH_PLACE=0 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fm 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag', d['ms_per_step'], d['stages_ms']['mtf'], d['stages_ms']['rle'], d['container']['ms_per_step_with_container'], d['container']['stages_ms'])"
done
cp /tmp/libtextcomp_orig.so text-compression_amd/libtextcomp.so
