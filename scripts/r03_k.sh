#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_msd.py tests/test_gpu_encode.py tests/test_gpu_fullsize.py tests/test_gpu_container_fused.py -x -q > gpurun_out/r03k_tests.log 2>&1
echo "tests rc=$?"; tail -n 8 gpurun_out/r03k_tests.log
timeout -k 10 300 python scripts/ab_env.py TC_SA_MSD_KEYONLY=0,1 2>&1 | tail -n 2
TC_BENCH_PLACE=0 timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-fm 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'], d['stages_ms'], d['container']['ms_per_step_with_container'])"
