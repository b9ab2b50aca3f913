#!/bin/bash
# round 3, first GPU call: the fused container path + the bench legs
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 420 python -m pytest tests/test_gpu_container_fused.py tests/test_gpu_container.py -x -q > gpurun_out/r03a_tests.log 2>&1
echo "tests rc=$?" | tee -a gpurun_out/r03a_tests.log
tail -n 15 gpurun_out/r03a_tests.log
timeout -k 10 400 python bench.py --steps 5 --warmup 1 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/r03a_bench.json; tail -n 5 gpurun_out/r03a_bench.err
