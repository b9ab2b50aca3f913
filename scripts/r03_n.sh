#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_msd.py tests/test_gpu_classes_digest.py -x -q > gpurun_out/r03n_tests.log 2>&1
echo "tests rc=$?"; tail -n 4 gpurun_out/r03n_tests.log
TC_SA_MSD_KEYONLY=0 timeout -k 10 300 python scripts/classes_bench.py 1073741824 acgt4,acgtn 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python scripts/classes_bench.py 1073741824 acgt4,acgtn,genome_like 2>&1 | grep -v amdgpu.ids
