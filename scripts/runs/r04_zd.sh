#!/bin/bash
# ranks in the first group pass when the LSD finish drowned in ties (many_ties): parity with the sample off (every text tries the
# finish pass first), the digest records, the long-period class
mkdir -p gpurun_out
TC_SA_SAMPLE=0 timeout -k 10 300 python tests/long/fuzz_long.py 400 251 120000 2>&1 | tail -n 1
TC_SA_SAMPLE=0 TC_SA_BIN_MIN_LOG2=0 TC_SA_SEG_MIN=1 timeout -k 10 300 python tests/long/fuzz_long.py 300 252 120000 2>&1 | tail -n 1
timeout -k 10 300 python tests/long/fuzz_chain.py 200 253 4000000 2>&1 | tail -n 1
timeout -k 10 600 python -m pytest tests/test_gpu_classes_digest.py tests/test_gpu_chain.py -x -q 2>&1 | tail -n 2
timeout -k 10 300 python scripts/classes_bench.py 1073741824 repeat_1MiB,repeat_4KiB,binary2,zipf_words 2>/dev/null | cut -c1-200
