#!/bin/bash
# equal-mass bins in the generic finish instances (4-letter DNA: the big instance; suffix arrays / index builds: the small one with values)
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_msd.py tests/test_gpu_encode.py tests/test_gpu_classes_digest.py tests/test_gpu_soak.py tests/test_gpu_fm_pairs.py tests/test_gpu_fmindex.py tests/test_gpu_fm_config4.py -x -q > gpurun_out/r03y_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -n 3 gpurun_out/r03y_tests.log
[ $rc -ne 0 ] && exit $rc
for v in 0 1; do for c in acgt4 acgtn; do TC_MSD_FINISH_LUT=$v timeout -k 10 200 python scripts/classes_bench.py 1073741824 $c 2>&1 | grep -E 'n=1073741824' | cut -c1-75 | sed "s/^/LUT=$v /"; done; done
