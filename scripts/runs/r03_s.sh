#!/bin/bash
# where does rle_nib_kernel spend its time?  (timing-only ablations: the containers of these runs are wrong)
for d in 0 1 2 3 4 6 7; do
  r=$(TC_RLE_DIAG=$d TC_BENCH_PLACE=0 timeout -k 10 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-fm 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['container'].get('stages_ms', d['container']))")
  echo "TC_RLE_DIAG=$d (1 no look-back B, 2 no strings, 4 no output): $r"
done
