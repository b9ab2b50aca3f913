#!/bin/bash
# step time against the number of workgroups of the MSD levels (TC_MSD_GRID), several processes per value,
# without the placement search: do slices that are not 2^25 bytes apart escape the slow mode?
for g in "$@"; do
  echo "== grid $g"
  for i in 1 2 3 4 5 6; do
    TC_BENCH_PLACE=0 TC_MSD_GRID=$g timeout -k 10 100 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   step %.2f ms  pass %.3f ms' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))" || exit 1
  done
done
