#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r03j_gputests.log 2>&1
echo "gpu tests rc=$?"; tail -n 4 gpurun_out/r03j_gputests.log
for d in 1 0 1 0; do
  TC_SA_DIRECT=$d TC_BENCH_PLACE=0 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fm 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DIRECT=$d', d['ms_per_step'], d['stages_ms'], d['container']['ms_per_step_with_container'])"
done
