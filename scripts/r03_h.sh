#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_msd.py tests/test_gpu_container_fused.py tests/test_gpu_fullsize.py tests/test_gpu_api_edges.py tests/test_gpu_mirror.py -x -q > gpurun_out/r03h_tests.log 2>&1
echo "tests rc=$?"; tail -n 5 gpurun_out/r03h_tests.log
for cfg in "1 1" "0 1" "1 0" "1 1"; do
  set -- $cfg
  TC_RLE_BLOCKED=$1 TC_MTF_SMALL=$2 TC_BENCH_PLACE=0 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fm 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('RLE_BLOCKED=$1 MTF_SMALL=$2', d['ms_per_step'], d['stages_ms']['mtf'], d['stages_ms']['rle'], d['container']['ms_per_step_with_container'], d['container']['stages_ms'])"
done
