#!/bin/bash
N=${1:-1073741824}
for b in 8 7 6 5 4; do
  TC_RADIX_VARIANT=0 TC_RADIX_DIGIT_BITS=$b timeout -k 10 200 python bench.py --n $N --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('bits $b: pass %.3f ms x %d  sa %.1f ms  step %.1f ms'%(d['roofline']['avg_launch_ms'], d['roofline']['launches_per_step'], d['stages_ms']['suffix_sort+bwt'], d['ms_per_step']))"
done
