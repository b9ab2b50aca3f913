#!/bin/bash
# parity suites, then in-process A/B of the selectors given as arguments, then the kernel table
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_msd.py tests/test_gpu_fullsize.py tests/test_gpu_api_edges.py tests/test_gpu_mirror.py tests/test_gpu_soak.py tests/test_gpu_classes_digest.py -x -q > gpurun_out/r03t_tests.log 2>&1
rc=$?
echo "tests rc=$rc"; tail -n 6 gpurun_out/r03t_tests.log
[ $rc -ne 0 ] && exit $rc
for spec in "$@"; do timeout -k 10 300 python scripts/ab_env.py $spec 2>&1 | tail -n 1; done
bash scripts/prof_brief.sh r03t --no-fm 2>&1 | grep -E "rle_|mtf_|finish"
