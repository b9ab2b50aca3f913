#!/bin/bash
# batched loads in group_kernel / rank_scatter_kernel: parity (encode, msd, soak slices, class digests), then the classes at 1 GiB
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_encode.py tests/test_gpu_msd.py tests/test_gpu_soak.py tests/test_gpu_classes_digest.py -x -q > gpurun_out/r03w_tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -n 3 gpurun_out/r03w_tests.log
[ $rc -ne 0 ] && exit $rc
for c in zipf_words genome_like runs_p0.9 binary_words binary2; do timeout -k 10 200 python scripts/classes_bench.py 1073741824 $c 2>&1 | grep -E 'n=1073741824' | cut -c1-110; done
