#!/bin/bash
# run bench.py with the given args, print a one-line summary
python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['stages_ms']
print('%.1f MB/s  step %.2f ms | sa %.2f mtf %.2f rle %.2f | pass %.3f ms x%d (%.0f GB/s, frac %.3f) | m=%s'%(d['value'],d['ms_per_step'],s['suffix_sort+bwt'],s['mtf'],s['rle'],r['avg_launch_ms'],r['launches_per_step'],r['achieved'],r['frac'],s['m']))"
