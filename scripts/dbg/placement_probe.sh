#!/bin/bash
# does the step time depend on where the buffers land?  N bench runs, buffer addresses + step time of each
for i in $(seq 1 ${1:-8}); do
  TC_SA_TRACE=2 timeout -k 10 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > /tmp/pp.out 2> /tmp/pp.err || exit 1
  grep -m1 "textcomp: buffers" /tmp/pp.err
  python -c "
import json,sys
d=json.loads([l for l in open('/tmp/pp.out') if l.startswith('{')][-1]); print('   step %.2f ms  pass %.3f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"
done
