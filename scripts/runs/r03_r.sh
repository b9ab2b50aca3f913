#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_encode.py tests/test_gpu_msd.py tests/test_gpu_container_fused.py tests/test_gpu_fullsize.py tests/test_gpu_api_edges.py tests/test_gpu_mirror.py tests/test_gpu_soak.py -x -q > gpurun_out/r03r_tests.log 2>&1
echo "tests rc=$?"; tail -n 4 gpurun_out/r03r_tests.log
timeout -k 10 300 python scripts/ab_env.py TC_RLE_BLOCKED=0,1 2>&1 | tail -n 1
bash scripts/prof_brief.sh r03r --no-fm 2>&1 | grep -E "rle_|mtf_"
