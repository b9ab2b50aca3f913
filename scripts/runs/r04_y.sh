#!/bin/bash
# finish_fix_kernel with its window loads issued together: parity (digest records, fuzz with copies / runs), then the class
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_classes_digest.py tests/test_gpu_encode.py -x -q > gpurun_out/r04y_digest.log 2>&1; echo "digest rc=$?"; tail -n 2 gpurun_out/r04y_digest.log
timeout -k 10 300 python tests/long/fuzz_long.py 300 77 120000 > gpurun_out/r04y_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -n 1 gpurun_out/r04y_fuzz.log
TC_SA_GLOBAL_PASSES=2 timeout -k 10 300 python tests/long/fuzz_long.py 200 78 120000 > gpurun_out/r04y_fuzz2.log 2>&1; echo "fuzz (2 global passes: long buckets) rc=$?"; tail -n 1 gpurun_out/r04y_fuzz2.log
timeout -k 10 300 python scripts/classes_bench.py 1073741824 genome_like,binary2 2>/dev/null | cut -c1-300
bash scripts/prof_class.sh genome_like 1073741824 2>&1 | grep -E "finish_fix|finish_kernel|table_build"
rm -rf gpurun_out/prof_cls_genome_like
