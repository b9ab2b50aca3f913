#!/bin/bash
# the container path with MTF + RLE + wire format in one kernel: parity suites, then timing
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_container_fused.py tests/test_gpu_comm.py tests/test_gpu_fullsize.py tests/test_gpu_soak.py tests/test_gpu_mirror.py tests/test_gpu_api_edges.py -x -q > gpurun_out/r03x_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -n 4 gpurun_out/r03x_tests.log
[ $rc -ne 0 ] && exit $rc
bash scripts/prof_brief.sh r03x --no-fm 2>&1 | grep -E "rle_|mtf_|checksum"
grep -o '"ms_per_step[a-z_]*": [0-9.]*' gpurun_out/prof_r03x_bench.log
