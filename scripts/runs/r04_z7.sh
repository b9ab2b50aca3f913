#!/bin/bash
# the inverse BWT's splitter offsets by a multiplicative hash: decode parity, the gaps record at 2^28, every class at 1 GiB
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_classes_digest.py tests/test_gpu_api_edges.py tests/test_gpu_fullsize.py -x -q > gpurun_out/r04z7_tests.log 2>&1; echo "tests rc=$?"; tail -n 2 gpurun_out/r04z7_tests.log
TC_IBWT_SEGCAP=64 timeout -k 10 300 python tests/long/fuzz_long.py 300 195 60000 > gpurun_out/r04z7_a.log 2>&1; echo "fuzz (tiny records: spare slots) rc=$?"; tail -n 1 gpurun_out/r04z7_a.log
TC_IBWT_LF=0 timeout -k 10 300 python tests/long/fuzz_long.py 200 196 60000 > gpurun_out/r04z7_b.log 2>&1; echo "fuzz (walk by positions) rc=$?"; tail -n 1 gpurun_out/r04z7_b.log
timeout -k 10 200 python scripts/dbg/dec_probe.py nrun 28 2>&1 | grep "^call" | tail -2
timeout -k 10 800 python scripts/classes_bench.py 1073741824 2>/dev/null | cut -c1-100
