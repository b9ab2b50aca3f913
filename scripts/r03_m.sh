#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_msd.py tests/test_gpu_encode.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r03m_tests.log 2>&1
echo "tests rc=$?"; tail -n 4 gpurun_out/r03m_tests.log
bash scripts/prof_brief.sh r03m --no-fm > gpurun_out/r03m_summary.txt 2>&1; head -n 8 gpurun_out/r03m_summary.txt
grep '^{' gpurun_out/prof_r03m_bench.log | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms'])"
