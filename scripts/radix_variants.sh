#!/bin/bash
# A/B of the radix-pass variants on the 1 GiB (or $1-byte) bench; prints per-pass ms.
N=${1:-1073741824}
for v in 0 1 4 2 3; do
  TC_RADIX_VARIANT=$v timeout -k 10 200 python bench.py --n $N --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('variant $v: %.1f MB/s  pass %.3f ms  sa %.1f ms  step %.1f ms'%(d['value'], d['roofline']['avg_launch_ms'], d['stages_ms']['suffix_sort+bwt'], d['ms_per_step']))"
done
